"""HIP kernels vs the CPU oracle, through the C ABI, on seeded inputs: bit exact."""
import numpy as np
import pytest
import torch

from aivc_amd import abi

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def eq(a_gpu, b_np):
    a = a_gpu.cpu().numpy()
    if isinstance(b_np, torch.Tensor):
        b_np = b_np.cpu().numpy()
    if a.dtype == np.int16 and b_np.dtype == np.uint16:
        a = a.view(np.uint16)
    if a.dtype == np.int32 and b_np.dtype == np.uint32:
        a = a.view(np.uint32)
    assert a.shape == b_np.shape
    np.testing.assert_array_equal(a, b_np)


CONV_CASES = [
    # mode, k, stride, pad, cin, cout, h, w, act1, act2, mul, res
    (abi.MODE_CONV, 5, 2, 2, 4, 8, 17, 23, 0, 0, False, False),
    (abi.MODE_CONV, 3, 1, 1, 8, 8, 9, 11, abi.ACT_LEAKY, 0, False, True),
    (abi.MODE_CONV, 3, 2, 1, 12, 16, 10, 14, abi.ACT_RELU, 0, False, False),
    (abi.MODE_CONV, 1, 2, 0, 8, 8, 9, 13, 0, 0, False, False),
    (abi.MODE_CONV, 1, 1, 0, 8, 8, 6, 10, abi.ACT_SIGMOID, 0, True, True),
    (abi.MODE_CONV, 3, 1, 1, 8, 8, 7, 9, 0, abi.ACT_RELU, False, True),
    (abi.MODE_TCONV, 5, 2, 0, 8, 6, 7, 9, 0, 0, False, False),
    (abi.MODE_TCONV, 3, 2, 0, 8, 8, 5, 6, abi.ACT_LEAKY, 0, False, True),
    (abi.MODE_TCONV, 5, 2, 0, 64, 3, 6, 5, 0, 0, False, False),
    (abi.MODE_GDN, 1, 1, 0, 8, 8, 5, 7, 0, 0, False, False),
    (abi.MODE_IGDN, 1, 1, 0, 16, 16, 5, 7, 0, 0, False, True),
    (abi.MODE_CONV, 5, 2, 2, 64, 128, 33, 47, 0, 0, False, False),
    (abi.MODE_CONV, 3, 1, 1, 128, 128, 19, 21, abi.ACT_LEAKY, 0, False, True),
    (abi.MODE_CONV, 3, 2, 1, 128, 128, 40, 44, abi.ACT_LEAKY, 0, False, False),
    (abi.MODE_CONV, 5, 2, 2, 12, 64, 47, 61, 0, 0, False, False),
    (abi.MODE_CONV, 5, 2, 2, 128, 64, 30, 34, 0, 0, False, False),
    (abi.MODE_CONV, 1, 1, 0, 128, 64, 23, 29, abi.ACT_LEAKY, 0, False, False),
    (abi.MODE_CONV, 1, 2, 0, 128, 128, 31, 29, 0, 0, False, False),
    (abi.MODE_TCONV, 5, 2, 0, 128, 128, 17, 19, 0, 0, False, False),
    (abi.MODE_TCONV, 3, 2, 0, 128, 128, 17, 19, abi.ACT_LEAKY, 0, False, True),
    (abi.MODE_TCONV, 5, 2, 0, 128, 64, 33, 35, 0, 0, False, False),
    (abi.MODE_TCONV, 5, 2, 0, 32, 128, 9, 11, abi.ACT_LEAKY, 0, False, False),
    (abi.MODE_GDN, 1, 1, 0, 128, 128, 33, 31, 0, 0, False, False),
    (abi.MODE_IGDN, 1, 1, 0, 64, 64, 33, 31, 0, 0, False, True),
    (abi.MODE_CONV, 3, 1, 1, 128, 192, 9, 11, 0, 0, False, False),
    (abi.MODE_CONV, 3, 1, 1, 64, 256, 20, 20, 0, 0, False, False),
    (abi.MODE_TCONV, 5, 2, 0, 64, 6, 37, 41, 0, 0, False, False),
    (abi.MODE_TCONV, 3, 2, 0, 64, 3, 17, 33, abi.ACT_LEAKY, 0, False, True),
    (abi.MODE_TCONV, 5, 2, 0, 128, 3, 9, 19, 0, 0, False, False),
    (abi.MODE_TCONV, 5, 2, 0, 16, 6, 16, 16, 0, abi.ACT_RELU, False, True),
    # thin outputs on the 16x16x4 MFMA kernel: partial tiles in x and y, several tiles, gate + residual
    (abi.MODE_TCONV, 5, 2, 0, 64, 3, 19, 70, abi.ACT_LEAKY, 0, True, False),
    (abi.MODE_TCONV, 3, 2, 0, 32, 6, 9, 40, 0, 0, False, False),
    (abi.MODE_TCONV, 5, 2, 0, 48, 6, 8, 33, 0, abi.ACT_LEAKY, True, True),
    (abi.MODE_TCONV, 3, 2, 0, 96, 3, 1, 1, 0, 0, False, False),
    (abi.MODE_TCONV, 5, 2, 0, 64, 6, 6, 10, abi.ACT_SIGMOID, 0, False, False),  # falls back to the VALU kernel
    (abi.MODE_TCONV, 5, 2, 0, 64, 3, 7, 70, abi.ACT_RELU, 0, False, False),     # lean epilogue, relu (+0.0 for negatives)
    (abi.MODE_TCONV, 5, 2, 0, 32, 6, 5, 33, abi.ACT_LEAKY, 0, False, False),
    # LDS-DMA K loop corner cases: a reduction of ONE K-tile (1x1, c_in 32), of an odd number (3x3 x 32 = 9), two
    # tiles; transposed with image-border zero fill on every tile; c_out beyond the tile width; rows beyond M
    (abi.MODE_CONV, 1, 1, 0, 32, 64, 9, 13, 0, 0, False, False),
    (abi.MODE_CONV, 3, 1, 1, 32, 32, 11, 7, abi.ACT_LEAKY, 0, False, True),
    (abi.MODE_CONV, 1, 2, 0, 64, 128, 5, 3, 0, abi.ACT_RELU, False, True),
    (abi.MODE_TCONV, 3, 2, 0, 32, 64, 7, 9, 0, 0, False, False),
    (abi.MODE_TCONV, 5, 2, 0, 64, 128, 1, 3, abi.ACT_LEAKY, 0, False, False),
    (abi.MODE_CONV, 5, 2, 2, 96, 160, 13, 9, 0, 0, False, False),
]


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('algo', [abi.ALGO_DIRECT, abi.ALGO_AUTO, abi.ALGO_MFMA])
def test_conv_family_bit_exact(case, algo, oracle, cuda):
    from aivc_amd import ops
    mode, k, s, pad, ci, co, h, w, a1, a2, use_mul, use_res = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((2, h, w, ci), dtype=np.float32)
    wt = (rng.standard_normal((co, k, k, ci), dtype=np.float32) / np.sqrt(k * k * ci)).astype(np.float32)
    bias = rng.standard_normal(co, dtype=np.float32)
    if mode in (abi.MODE_GDN, abi.MODE_IGDN):
        wt = np.abs(wt) * 0.1
        bias = np.abs(bias) + 0.1
    ho, wo = abi.conv_out_size(mode, h, w, k, s, pad)
    mul = rng.standard_normal((2, ho, wo, co), dtype=np.float32) if use_mul else None
    res = rng.standard_normal((2, ho, wo, co), dtype=np.float32) if use_res else None
    ref = oracle.conv2d(x, wt, bias, mode=mode, stride=s, pad=pad, act1=a1, act2=a2, mul=mul, res=res)
    got = ops.conv2d(T(x, cuda), T(wt, cuda), T(bias, cuda), mode=mode, stride=s, pad=pad, act1=a1, act2=a2,
                     mul=None if mul is None else T(mul, cuda), res=None if res is None else T(res, cuda),
                     algo=algo)
    eq(got, ref)


@pytest.mark.parametrize('grid', [1, 3, 7])
@pytest.mark.parametrize('co,k,ci,h,w', [(3, 5, 64, 21, 100), (6, 5, 64, 9, 70), (3, 3, 16, 13, 65)])
def test_thin_layer_tile_walk(grid, co, k, ci, h, w, oracle, cuda, monkeypatch):
    """The thin output layer's persistent groups walk several tiles each (origins advanced without divisions across tile
    rows and images, double-buffered patches, epilogue of the previous tile inside the next chain): forced here at
    small sizes with AIVC_THIN_GRID_MAX groups over 3 images, bit exact against the oracle -- with and without bias."""
    from aivc_amd import ops
    monkeypatch.setenv('AIVC_THIN_GRID_MAX', str(grid))
    rng = np.random.default_rng(grid * 100 + co)
    x = rng.standard_normal((3, h, w, ci), dtype=np.float32)
    wt = (rng.standard_normal((co, k, k, ci), dtype=np.float32) / np.sqrt(k * k * ci)).astype(np.float32)
    bias = rng.standard_normal(co, dtype=np.float32)
    for b in (bias, None):
        ref = oracle.conv2d(x, wt, b, mode=abi.MODE_TCONV, stride=2, pad=0, act1=abi.ACT_LEAKY)
        got = ops.conv2d(T(x, cuda), T(wt, cuda), None if b is None else T(b, cuda), mode=abi.MODE_TCONV, stride=2, pad=0,
                         act1=abi.ACT_LEAKY)
        eq(got, ref)


def test_frame_ops_bit_exact(oracle, cuda):
    from aivc_amd import ops
    rng = np.random.default_rng(5)
    for (h, w) in [(9, 13), (10, 14), (16, 16), (1, 1), (6, 1028)]:
        hc, wc = (h + 1) // 2, (w + 1) // 2
        y8 = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
        u8 = rng.integers(0, 256, (2, hc, wc), dtype=np.uint8)
        v8 = rng.integers(0, 256, (2, hc, wc), dtype=np.uint8)
        eq(ops.yuv420_to_444(T(y8, cuda), T(u8, cuda), T(v8, cuda), c_store=4),
           oracle.yuv420u8_to_444(y8, u8, v8, c_store=4))
        yf, uf, vf = (a.astype(np.float32) / np.float32(255) for a in (y8, u8, v8))
        eq(ops.yuv420_to_444(T(yf, cuda), T(uf, cuda), T(vf, cuda), c_store=4),
           oracle.yuv420_to_444(yf, uf, vf, c_store=4))
        x = (rng.standard_normal((2, h + 3, w + 2, 4), dtype=np.float32) * 0.4 + 0.5).astype(np.float32)
        skip = (rng.standard_normal((2, h, w, 4), dtype=np.float32) * 0.1).astype(np.float32)
        for sk in (None, skip):
            rf, rb = oracle.frame_to_yuv420(x, h, w, skip=sk)
            gf, gb = ops.frame_to_yuv420(T(x, cuda), h, w, skip=None if sk is None else T(sk, cuda))
            for a, b in zip(gf + gb, rf + rb):
                eq(a, b)
        # the synthesis output as the codec hands it over: 3 channels, padded to an even row length (the 8-byte /
        # 16-byte load path of the kernel when the frame sides are even, the scalar one otherwise)
        x3 = np.ascontiguousarray(x[:, :, :w + 2 - (w & 1), :3])
        for sk in (None, skip):
            rf, rb = oracle.frame_to_yuv420(x3, h, w, skip=sk)
            gf, gb = ops.frame_to_yuv420(T(x3, cuda), h, w, skip=None if sk is None else T(sk, cuda))
            for a, b in zip(gf + gb, rf + rb):
                eq(a, b)
            _, gb8 = ops.frame_to_yuv420(T(x3, cuda), h, w, skip=None if sk is None else T(sk, cuda), want_float=False)
            for a, b in zip(gb8, rb):
                eq(a, b)


def test_warp_bit_exact(oracle, cuda):
    from aivc_amd import ops
    rng = np.random.default_rng(6)
    for (h, w, s) in [(9, 13, 3.0), (8, 8, 30.0), (5, 1, 2.0), (32, 48, 1.0)]:
        x = rng.standard_normal((2, h, w, 4), dtype=np.float32)
        flow = (rng.standard_normal((2, h, w, 2), dtype=np.float32) * s).astype(np.float32)
        eq(ops.warp(T(x, cuda), T(flow, cuda)), oracle.warp(x, flow))
        mof = (rng.standard_normal((2, h + 2, w + 1, 8), dtype=np.float32)).astype(np.float32)
        mof[..., 2:6] *= s
        prev = rng.random((2, h, w, 4), dtype=np.float32)
        nxt = rng.random((2, h, w, 4), dtype=np.float32)
        for ft in (1, 2):
            r = oracle.warp_blend(mof, prev, nxt, h, w, ft)
            g = ops.warp_blend(T(mof, cuda), T(prev, cuda), T(nxt, cuda), h, w, ft, want_aux=True)
            for kk in ('pred', 'skip', 'x_warp', 'alpha', 'beta'):
                eq(g[kk], r[kk])
            # the general kernel (3 stored reference channels, 3 output channels, 7 mask / flow channels) against
            # the 16-byte fast path above (4 / 4 / 8)
            mof7 = np.ascontiguousarray(mof[..., :7])
            r3 = oracle.warp_blend(mof7, prev[..., :3].copy(), nxt[..., :3].copy(), h, w, ft, co=3)
            g3 = ops.warp_blend(T(mof7, cuda), T(prev[..., :3].copy(), cuda), T(nxt[..., :3].copy(), cuda), h, w, ft, co=3,
                                want_aux=True)
            for kk in ('pred', 'skip', 'x_warp', 'alpha', 'beta'):
                eq(g3[kk], r3[kk])
                if kk in ('pred', 'skip', 'x_warp'):
                    np.testing.assert_array_equal(r3[kk], r[kk][..., :3])


def test_latent_ops_bit_exact(oracle, cuda):
    from aivc_amd import ops
    rng = np.random.default_rng(7)
    hs = (rng.standard_normal((1, 6, 9, 16), dtype=np.float32) * 8).astype(np.float32)
    hs[0, 0, 0, 8], hs[0, 0, 1, 8] = -30, 30
    mu, sg = oracle.hyper_params(hs, 8, 5, 7)
    gmu, gsg = ops.hyper_params(T(hs, cuda), 8, 5, 7)
    eq(gmu, mu)
    eq(gsg, sg)
    y = (rng.standard_normal((1, 5, 7, 8), dtype=np.float32) * 20).astype(np.float32)
    y[0, 0, 0, :4] = [0.5, 1.5, 2.5, -0.5]
    y[0, 0, 1, :2] = [400, -400]
    gain = rng.standard_normal(8).astype(np.float32)
    eq(ops.channel_gain(T(y, cuda), T(gain, cuda)), oracle.channel_gain(y, gain))
    q, yh = oracle.quantize_center(y, mu, gain)
    gq, gyh = ops.quantize_center(T(y, cuda), T(mu, cuda), T(gain, cuda))
    eq(gq, q)
    eq(gyh, yh)
    q0, yh0 = oracle.quantize_center(y)
    gq0, gyh0 = ops.quantize_center(T(y, cuda))
    eq(gq0, q0)
    eq(gyh0, yh0)
    eq(ops.dequantize(T(q, cuda), T(mu, cuda), T(gain, cuda)), oracle.dequantize(q, mu, gain))
    beta = np.abs(rng.standard_normal(8)).astype(np.float32)
    gamma = (rng.standard_normal((8, 8)) * 0.1).astype(np.float32)
    be, ge = oracle.gdn_reparam(beta, gamma, 1e-3, 2 ** -18, 2 ** -36)
    gbe, gge = ops.gdn_reparam(T(beta, cuda), T(gamma, cuda), 1e-3, 2 ** -18, 2 ** -36)
    eq(gbe, be)
    eq(gge, ge)


def test_cdf_kernels_bit_exact(oracle, cuda):
    from aivc_amd import ops
    rng = np.random.default_rng(8)
    params = (rng.standard_normal((6, abi.BALLE_PARAMS)) * 1.2).astype(np.float32)
    table, cdf = oracle.balle_cdf_table(params)
    gt, gc = ops.balle_cdf_table(T(params, cuda), want_float=True)
    eq(gc, cdf)
    eq(gt, table)
    sig = np.exp(rng.uniform(np.log(1e-4), np.log(148.4), (1, 6, 7, 8))).astype(np.float32)
    sig[0, 0, 0, 0], sig[0, 0, 0, 1] = 1e-4, 148.41316
    maps = [0, 2, 3, 7]
    eq(ops.laplace_cdf_rows(T(sig, cuda), maps), oracle.laplace_cdf_rows(sig, maps))
    q = np.clip(np.rint(rng.laplace(0, 1, sig.shape) * sig), -256, 255).astype(np.int16)
    eq(ops.laplace_bounds(T(sig, cuda), T(q, cuda), maps), oracle.laplace_bounds(sig, q, maps))
    qz = rng.integers(-5, 6, (1, 3, 4, 6)).astype(np.int16)
    eq(ops.table_bounds(T(table.view(np.int16), cuda), T(qz, cuda)), oracle.table_bounds(table, qz))
    flags = ops.nonzero_flags(T(q, cuda)).cpu().numpy()[0]
    assert [i for i in range(8) if flags[i]] == oracle.nonzero_maps(q)
    # the frames of a batch in one launch: every image has its own set of all-zero maps
    qb = np.clip(np.rint(rng.laplace(0, 1, (5, 6, 7, 8)) * 3), -256, 255).astype(np.int16)
    for i, dead in enumerate(([], [0], [1, 7], list(range(8)), [3])):
        qb[i][..., dead] = 0
    fb = ops.nonzero_flags(T(qb, cuda)).cpu().numpy()
    for i in range(5):
        assert [k for k in range(8) if fb[i][k]] == oracle.nonzero_maps(qb[i:i + 1])


@pytest.mark.parametrize('n_sym,scale', [(1, 1.0), (63, 0.3), (64, 2.0), (65, 5.0), (1000, 0.05), (5000, 1.0),
                                         (20000, 40.0), (3000, 1e-4), (70000, 0.8)])
def test_range_coder_bit_exact(n_sym, scale, oracle, cuda):
    """(70000 symbols: the encoder launch that asks for a CU of its own, csrc/entropy.hip aivc_range_encode)"""
    from aivc_amd import ops
    rng = np.random.default_rng(n_sym)
    c = 4
    npix = (n_sym + c - 1) // c
    sig = (np.exp(rng.uniform(np.log(0.05), np.log(4.0), (1, 1, npix, c))) * scale).astype(np.float32)
    sig = np.clip(sig, 1e-4, 148.4).astype(np.float32)
    q = np.clip(np.rint(rng.laplace(0, 1, sig.shape) * sig / np.sqrt(2)), -256, 255).astype(np.int16)
    maps = list(range(c))
    bounds = oracle.laplace_bounds(sig, q, maps)[:n_sym]
    ref_bytes = oracle.range_encode(bounds)
    out, lens, offs = ops.range_encode([T(bounds.view(np.int32), cuda)])
    ln = int(lens.cpu()[0])
    got_bytes = out.cpu().numpy()[:ln].tobytes()
    assert got_bytes == ref_bytes
    rows = oracle.laplace_cdf_rows(sig, maps)[:n_sym]
    ref_sym = oracle.range_decode(ref_bytes, rows, n_sym)
    want = (q.reshape(-1, c).T.reshape(-1)[:n_sym].astype(np.int32) + 256).astype(np.uint16)
    np.testing.assert_array_equal(ref_sym, want)
    got_sym, bits = ops.range_decode([ref_bytes], T(rows.view(np.int16), cuda), [0], [n_sym], [0], want_bits=True)
    eq(got_sym[0], ref_sym)
    # bits shifted in by renormalisation: same count as the oracle's, and it accounts for the payload length
    _, ref_bits = oracle.range_decode(ref_bytes, rows, n_sym, want_bits=True)
    assert int(bits.cpu()[0]) == ref_bits and len(ref_bytes) == (ref_bits + 2 + 7) // 8


def _straddle_stream(rng, n, burst):
    """packed (c_lo | c_hi << 16) bounds whose intervals keep sitting across the middle of the coder's range: every such
    symbol adds ~14 straddle (E3) steps to the pending count, `burst` of them in a row push it past 32 and far beyond,
    then a symbol that settles releases the run -- the encoder's long-run path, followed by ordinary symbols"""
    out = []
    while len(out) < n:
        for _ in range(int(rng.integers(1, burst + 1))):
            d = int(rng.integers(1, 4))
            out.append((0x8000 - d) | ((0x8000 + int(rng.integers(1, 4))) << 16))
        lo = int(rng.integers(0, 0xF000))
        out.append(lo | ((lo + int(rng.integers(1, 0x0FFF))) << 16))
        for _ in range(int(rng.integers(0, 40))):
            lo = int(rng.integers(0, 0xFFF0))
            hi = lo + int(rng.integers(1, 0x10000 - lo))
            out.append(lo | ((hi & 0xFFFF) << 16))  # hi = 2^16 packs as 0
    return np.array(out[:n], np.uint32)


@pytest.mark.parametrize('kernel', ['lanes', 'wave'])
def test_range_encoder_batches_ragged_streams_and_long_straddle_runs(kernel, oracle, cuda, monkeypatch):
    """A batch of streams in ONE launch (the stream-per-lane encoder codes them side by side in one wavefront, the
    wave-per-stream encoder one wavefront each): ragged lengths incl. empty, 1, 7, 8, 9 symbols (chunk edges), more than
    64 streams (two launches), ordinary Laplace symbols and adversarial straddle runs (pending count up to hundreds: the
    output of one symbol is longer than a word) -- every stream's bytes == the oracle's."""
    from aivc_amd import ops
    monkeypatch.setenv('AIVC_RC_ENCODE', kernel)
    rng = np.random.default_rng(77)
    lens = [0, 1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 500, 2049] + [int(v) for v in rng.integers(1, 3000, 60)]
    streams = []
    for i, n in enumerate(lens):
        if i % 3 == 2:
            streams.append(_straddle_stream(rng, n, burst=1 + i % 9))
        else:
            sig = np.clip(np.exp(rng.uniform(np.log(0.05), np.log(40.0), (1, 1, max(n, 1), 1))), 1e-4, 148.4).astype(np.float32)
            q = np.clip(np.rint(rng.laplace(0, 1, sig.shape) * sig / np.sqrt(2)), -256, 255).astype(np.int16)
            streams.append(oracle.laplace_bounds(sig, q, [0])[:n])
    want = [oracle.range_encode(b) for b in streams]
    assert max(len(w) for w in want) > 0
    out, ln, offs = ops.range_encode([T(np.ascontiguousarray(b).view(np.int32), cuda) for b in streams])
    out_h, ln_h = out.cpu().numpy(), ln.cpu().numpy()
    for i, ((off, cap), n, w) in enumerate(zip(offs, ln_h, want)):
        assert out_h[off:off + int(n)].tobytes() == w, (kernel, i, lens[i])


@pytest.mark.parametrize('kernel', ['lanes', 'wave'])
def test_range_encoder_output_capacity_contract(kernel, oracle, cuda, monkeypatch):
    """include/aivc_hip.h aivc_rc_stream.out_cap: whole 32-bit words, at least one -- anything else is AIVC_ERR_ARG at the
    entry point (the stream-per-lane packer clamps its stores to word out_cap / 4 - 1); a capacity the stream does not
    fit reports out_len 0xFFFFFFFF and writes nothing beyond it"""
    import ctypes as C
    from aivc_amd import abi, ops
    from aivc_amd._lib import AivcNativeError, call
    monkeypatch.setenv('AIVC_RC_ENCODE', kernel)
    rng = np.random.default_rng(5)
    bounds = _straddle_stream(rng, 400, burst=3)
    want = oracle.range_encode(bounds)
    assert len(want) > 64
    b = T(np.ascontiguousarray(bounds).view(np.int32), cuda)
    guard = 0xA5
    for cap, ok in ((0, False), (3, False), (6, False), (8, True), (64, True), ((len(want) + 3) // 4 * 4, True)):
        out = torch.full((4096,), guard, dtype=torch.uint8, device=cuda)
        ln = torch.zeros(1, dtype=torch.int32, device=cuda)
        batch = abi.RcBatch()
        batch.n_streams = 1
        batch.s[0].in_off, batch.s[0].out_off, batch.s[0].n_sym, batch.s[0].out_cap = 0, 0, b.numel(), cap
        args = ('aivc_range_encode', C.c_void_p(b.data_ptr()), C.byref(batch), C.c_void_p(out.data_ptr()),
                C.c_void_p(ln.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if not ok:
            with pytest.raises(AivcNativeError):
                call(*args)
            continue
        call(*args)
        torch.cuda.synchronize()
        n = int(ln.cpu().numpy().view(np.uint32)[0])
        out_h = out.cpu().numpy()
        assert (out_h[cap:] == guard).all(), (kernel, cap)  # nothing beyond the capacity
        if cap >= len(want):
            assert n == len(want) and out_h[:n].tobytes() == want
        else:
            assert n == 0xFFFFFFFF, (kernel, cap, n)


@pytest.mark.parametrize('scale', [0.4, 3.0, 40.0, 150.0])
def test_range_decode_from_windows(scale, oracle, cuda):
    """64-entry CDF windows + sigma per position (what the codec's decoder reads) give the symbols of the full rows,
    in and outside the window; the windows equal the oracle's and the slice of the full rows"""
    from aivc_amd import ops
    rng = np.random.default_rng(int(scale * 10))
    h, w, c = 9, 13, 6
    maps = [0, 2, 5]
    sig = (np.abs(rng.standard_normal((1, h, w, c))) * scale + 0.05).astype(np.float32)
    q = np.clip(np.rint(rng.standard_normal((1, h, w, c)) * sig), -256, 255).astype(np.int16)
    bounds = oracle.laplace_bounds(sig, q, maps)
    payload = oracle.range_encode(bounds)
    rows = oracle.laplace_cdf_rows(sig, maps)
    n_sym = len(maps) * h * w
    want = oracle.range_decode(payload, rows, n_sym)
    win_o, sp_o = oracle.laplace_cdf_windows(sig, maps)
    np.testing.assert_array_equal(win_o, rows[:, abi.CDF_WIN0:abi.CDF_WIN0 + abi.CDF_WIN])
    np.testing.assert_array_equal(oracle.range_decode_windows(payload, win_o, sp_o, n_sym), want)
    win, sp = ops.laplace_cdf_windows(T(sig, cuda), maps)
    eq(win, win_o)
    eq(sp, sp_o)
    got, bits = ops.range_decode([payload], win, [0], [n_sym], [0], sigma_pos=sp, want_bits=True)
    eq(got[0], want)
    assert len(payload) == (int(bits.cpu()[0]) + 2 + 7) // 8
    if scale >= 40.0:
        assert (np.abs(q[..., maps]) > 32).any(), 'the slow path (symbol outside the window) must be exercised'


@pytest.mark.timeout(120)
@pytest.mark.parametrize('sigma', [0.7, 30.0, 148.0, 600.0])
def test_range_decode_from_windows_of_a_foreign_stream(sigma, oracle, cuda):
    """Bytes no encoder of these CDFs wrote (a corrupt or foreign stream): the decoder's rare path must end its search
    whatever the stream asks for -- at large sigma entry 0 of a row is not 0 and the stream can ask for less -- and
    return what the kernel that searches full rows returns (and the oracle, where every count has a symbol: once a
    stream asks for less than entry 0 the interval is invalid and torchac's own arithmetic is undefined)."""
    from aivc_amd import ops
    rng = np.random.default_rng(int(sigma * 10))
    n_sym = 3000
    sig = np.full((1, 1, n_sym, 1), sigma, np.float32)
    payload = bytes(rng.integers(0, 256, 6000, dtype=np.uint8)) if sigma != 30.0 else bytes(6000)
    rows = oracle.laplace_cdf_rows(sig, [0])
    want = oracle.range_decode(payload, rows, n_sym)
    win, sp = ops.laplace_cdf_windows(T(sig, cuda), [0])
    got_w = ops.range_decode([payload], win, [0], [n_sym], [0], sigma_pos=sp)[0]
    got_r = ops.range_decode([payload], ops.laplace_cdf_rows(T(sig, cuda), [0]), [0], [n_sym], [0])[0]
    assert torch.equal(got_w, got_r)
    if int(rows[0, 0]) == 0:
        eq(got_w, want)


@pytest.mark.parametrize('sigma', [1e-4, 0.3, 5.0, 148.0])
def test_range_coder_forced_symbols(sigma, oracle, cuda):
    """Deterministic visit of the decoder's corners: symbols 0, 1 (octet 0), 223 / 224 and 287 / 288 (either side of
    the 64-entry window), 286, 510, 511 (octet 63: reads CDF entry 512 through lane 8) and 512 (value +256, torchac's
    max_symbol, upper bound 2^16 packed as c_hi = 0) -- HIP bounds / bytes / symbols == oracle, from full rows and
    from windows, at a sigma where everything sits in the window's tail and one where nothing does."""
    from aivc_amd import ops
    from test_rangecoder import forced_case
    sig, q = forced_case(sigma, repeat=11)
    n = q.size
    want = (q.reshape(-1).astype(np.int32) + 256).astype(np.uint16)
    b_ref = oracle.laplace_bounds(sig, q, [0])
    b = ops.laplace_bounds(T(sig, cuda), T(q, cuda), [0])
    eq(b, b_ref.view(np.int32))
    ref_bytes = oracle.range_encode(b_ref)
    out, lens, _ = ops.range_encode([b])
    assert out.cpu().numpy()[:int(lens.cpu()[0])].tobytes() == ref_bytes
    rows = ops.laplace_cdf_rows(T(sig, cuda), [0])
    eq(rows, oracle.laplace_cdf_rows(sig, [0]).view(np.int16))
    eq(ops.range_decode([ref_bytes], rows, [0], [n], [0])[0], want)
    win, sp = ops.laplace_cdf_windows(T(sig, cuda), [0])
    eq(ops.range_decode([ref_bytes], win, [0], [n], [0], sigma_pos=sp)[0], want)


def test_range_coder_symbol_512_table_mode(oracle, cuda):
    from aivc_amd import ops
    from test_rangecoder import FORCED
    rng = np.random.default_rng(5)
    params = (rng.standard_normal((2, abi.BALLE_PARAMS)) * 0.8).astype(np.float32)
    table, _ = oracle.balle_cdf_table(params)
    qz = np.array(FORCED * 2, np.int16).reshape(1, 2, 5, 2)
    b_ref = oracle.table_bounds(table, qz)
    b = ops.table_bounds(T(table.view(np.int16), cuda), T(qz, cuda))
    eq(b, b_ref.view(np.int32))
    ref_bytes = oracle.range_encode(b_ref)
    out, lens, _ = ops.range_encode([b])
    assert out.cpu().numpy()[:int(lens.cpu()[0])].tobytes() == ref_bytes
    sym = ops.range_decode([ref_bytes], T(table.view(np.int16), cuda), [0], [qz.size], [10])[0]
    eq(ops.scatter_symbols(sym, 10, 2, [0, 1]), qz.reshape(10, 2))


def test_range_coder_pmf_and_scatter(oracle, cuda):
    from aivc_amd import ops
    rng = np.random.default_rng(11)
    params = (rng.standard_normal((5, abi.BALLE_PARAMS)) * 0.8).astype(np.float32)
    table, _ = oracle.balle_cdf_table(params)
    qz = rng.integers(-3, 4, (1, 6, 7, 5)).astype(np.int16)
    bounds = oracle.table_bounds(table, qz)
    ref_bytes = oracle.range_encode(bounds)
    out, lens, _ = ops.range_encode([T(bounds.view(np.int32), cuda)])
    assert out.cpu().numpy()[:int(lens.cpu()[0])].tobytes() == ref_bytes
    n_sym = qz.size
    sym = ops.range_decode([ref_bytes], T(table.view(np.int16), cuda), [0], [n_sym], [42])[0]
    eq(sym, oracle.range_decode(ref_bytes, table, n_sym, plane=42))
    qback = ops.scatter_symbols(sym, 42, 5, list(range(5)))
    eq(qback, qz.reshape(42, 5))
    # partial map list
    maps = [1, 4]
    s2 = T((qz.reshape(42, 5)[:, maps].T.reshape(-1).astype(np.int32) + 256).astype(np.uint16).view(np.int16), cuda)
    eq(ops.scatter_symbols(s2, 42, 5, maps), oracle.scatter_symbols(s2.cpu().numpy().view(np.uint16), 42, 5, maps))


def test_range_coder_many_streams_concurrently(oracle, cuda):
    """70 independent streams (> 64 per launch) with different lengths: batched == one by one."""
    from aivc_amd import ops
    rng = np.random.default_rng(21)
    c, bl, pl, rows_all, offs, ns, want = 1, [], [], [], [], [], []
    total = 0
    for i in range(70):
        n = int(rng.integers(1, 700))
        sig = np.exp(rng.uniform(-3, 3, (1, 1, n, 1))).clip(1e-4, 148).astype(np.float32)
        q = np.clip(np.rint(rng.laplace(0, 1, sig.shape) * sig), -256, 255).astype(np.int16)
        b = oracle.laplace_bounds(sig, q, [0])
        bl.append(T(b.view(np.int32), cuda))
        pl.append(oracle.range_encode(b))
        rows_all.append(oracle.laplace_cdf_rows(sig, [0]))
        offs.append(total)
        total += n
        ns.append(n)
        want.append((q.reshape(-1).astype(np.int32) + 256).astype(np.uint16))
    out, lens, o = ops.range_encode(bl)
    out_h, lens_h = out.cpu().numpy(), lens.cpu().numpy()
    for i in range(70):
        assert out_h[o[i][0]:o[i][0] + lens_h[i]].tobytes() == pl[i]
    rows = T(np.concatenate(rows_all).view(np.int16), cuda)
    dec = ops.range_decode(pl, rows, offs, ns, [0] * 70)
    for i in range(70):
        eq(dec[i], want[i])


@pytest.mark.parametrize('case', [
    # mode, k, stride, pad, cin, cout, h, w, inverse, res
    (abi.MODE_CONV, 5, 2, 2, 12, 64, 31, 45, False, False),
    (abi.MODE_CONV, 5, 2, 2, 64, 128, 33, 29, False, False),
    (abi.MODE_CONV, 3, 1, 1, 128, 128, 17, 19, False, True),
    (abi.MODE_CONV, 3, 1, 1, 128, 128, 17, 19, True, True),
    (abi.MODE_TCONV, 5, 2, 0, 128, 128, 9, 11, True, False),
    (abi.MODE_TCONV, 5, 2, 0, 128, 64, 23, 21, True, False),
    (abi.MODE_CONV, 3, 1, 1, 32, 32, 9, 9, False, False),
    (abi.MODE_CONV, 3, 1, 1, 8, 8, 9, 9, False, True),      # not fusable: two launches
])
def test_fused_gdn_bit_exact(case, oracle, cuda):
    """conv + (I)GDN fused in one launch == oracle fused == oracle conv followed by oracle GDN"""
    from aivc_amd import ops
    mode, k, s, pad, ci, co, h, w, inv, use_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((2, h, w, ci), dtype=np.float32)
    wt = (rng.standard_normal((co, k, k, ci), dtype=np.float32) / np.sqrt(k * k * ci)).astype(np.float32)
    bias = rng.standard_normal(co, dtype=np.float32)
    beta = (np.abs(rng.standard_normal(co)) + 0.2).astype(np.float32)
    gamma = (np.abs(rng.standard_normal((co, co))) * 0.05).astype(np.float32)
    ho, wo = abi.conv_out_size(mode, h, w, k, s, pad)
    res = rng.standard_normal((2, ho, wo, co), dtype=np.float32) if use_res else None
    two = oracle.gdn(oracle.conv2d(x, wt, bias, mode=mode, stride=s, pad=pad), beta, gamma, inverse=inv, res=res)
    fused = oracle.conv2d(x, wt, bias, mode=mode, stride=s, pad=pad, res=res, gdn=(beta, gamma, inv))
    np.testing.assert_array_equal(fused, two)
    got = ops.conv2d(T(x, cuda), T(wt, cuda), T(bias, cuda), mode=mode, stride=s, pad=pad,
                     res=None if res is None else T(res, cuda), gdn=(T(beta, cuda), T(gamma, cuda), inv))
    eq(got, two)


def test_gdn_lean_math_selfcheck(cuda):
    """the lean square root / division of the fused GDN epilogues (csrc/common.h) == the compiler's IEEE sequences:
    every float of the safe range for the square root, 2^39 random operand pairs (two seeds) for the division"""
    from aivc_amd import ops
    for seed in (20260929, 7):
        bad_sqrt, bad_div = ops.selfcheck_gdn_math(1 << 38, seed=seed)
        assert (bad_sqrt, bad_div) == (0, 0), seed


@pytest.mark.parametrize('scale,zero_bias', [(1.0, False), (2.0 ** 70, False), (2.0 ** -70, True), (0.0, True), (2.0 ** 40, False)])
@pytest.mark.parametrize('inv', [False, True])
def test_fused_gdn_operand_range_fallback(scale, zero_bias, inv, oracle, cuda):
    """operands far outside [2^-60, 2^60] (huge, tiny, exactly zero outputs) next to ordinary ones: the image layer
    (5x5 s2 -> 64, aivc_conv_images) takes the full IEEE sequences for the wavefronts that see them and the lean ones
    (csrc/common.h) for the others; the 3x3 128 -> 128 layer always the full ones: either way == oracle, bit for bit"""
    from aivc_amd import ops
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((2, 16, 32, 128), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    x[0, :8] *= np.float32(1.0 if scale >= 1 else 2.0 ** 60)  # mixed: some tiles in range, some not
    wt = (rng.standard_normal((128, 3, 3, 128), dtype=np.float32) / 34.0).astype(np.float32)
    bias = np.zeros(128, np.float32) if zero_bias else rng.standard_normal(128, dtype=np.float32)
    beta = (np.abs(rng.standard_normal(128)) + 0.2).astype(np.float32)
    gamma = (np.abs(rng.standard_normal((128, 128))) * 0.05).astype(np.float32)
    with np.errstate(over='ignore', invalid='ignore'):
        want = oracle.conv2d(x, wt, bias, stride=1, pad=1, gdn=(beta, gamma, inv))
    got = ops.conv2d(T(x, cuda), T(wt, cuda), T(bias, cuda), stride=1, pad=1, gdn=(T(beta, cuda), T(gamma, cuda), inv))
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # the first analysis layer from float sources (aivc_conv_images)
    img = (rng.random((2, 16, 128, 4), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    img[..., 3] = 0
    wi = np.zeros((64, 5, 5, 4), np.float32)
    wi[..., :3] = rng.standard_normal((64, 5, 5, 3), dtype=np.float32) * 0.1
    bi = np.zeros(64, np.float32) if zero_bias else rng.standard_normal(64, dtype=np.float32)
    b64, g64 = beta[:64].copy(), gamma[:64, :64].copy()
    with np.errstate(over='ignore', invalid='ignore'):
        want_i = oracle.conv2d(img, wi, bi, stride=2, pad=2, gdn=(b64, g64, inv))
    stack = ops.ImageStack([T(img, cuda)], 16, 128, cuda)
    got_i = ops.conv2d(stack, T(wi, cuda), T(bi, cuda), stride=2, pad=2, gdn=(T(b64, cuda), T(g64, cuda), inv))
    np.testing.assert_array_equal(got_i.cpu().numpy().view(np.uint32), want_i.view(np.uint32))


@pytest.mark.parametrize('case', [
    # k, stride, cin, c_mid, c_tail, n, h, w, act1, act2, res
    (3, 1, 64, 64, 128, 2, 16, 32, abi.ACT_LEAKY, abi.ACT_LEAKY, True),   # whole 128-pixel tiles (the bottleneck block)
    (3, 1, 64, 64, 128, 2, 17, 19, abi.ACT_LEAKY, abi.ACT_LEAKY, True),   # ragged last tile
    (3, 1, 64, 64, 128, 1, 9, 5, abi.ACT_RELU, abi.ACT_NONE, True),       # a single partial tile
    (3, 1, 64, 64, 128, 2, 13, 21, abi.ACT_NONE, abi.ACT_RELU, False),
    (5, 2, 32, 64, 128, 2, 31, 27, abi.ACT_LEAKY, abi.ACT_NONE, False),
    (1, 1, 128, 64, 128, 3, 11, 23, abi.ACT_RELU, abi.ACT_LEAKY, True),
    (3, 1, 64, 64, 64, 2, 9, 9, abi.ACT_LEAKY, abi.ACT_LEAKY, True),      # not fusable (tail width): two launches
    (3, 1, 8, 12, 24, 2, 9, 9, abi.ACT_LEAKY, abi.ACT_LEAKY, True),       # not fusable (narrow): two launches
    (3, 1, 12, 6, 12, 1, 7, 9, abi.ACT_LEAKY, abi.ACT_LEAKY, True),       # intermediate width not a multiple of 4
])
def test_fused_tail_bit_exact(case, oracle, cuda):
    """conv + activation + 1x1 conv (+ residual, activation) in one launch == the oracle's two convolutions"""
    from aivc_amd import ops
    k, s, ci, cm, ct, n, h, w, a1, a2, use_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((n, h, w, ci), dtype=np.float32)
    wt = (rng.standard_normal((cm, k, k, ci), dtype=np.float32) / np.sqrt(k * k * ci)).astype(np.float32)
    b1 = rng.standard_normal(cm, dtype=np.float32)
    cm4 = (cm + 3) // 4 * 4
    w3 = np.zeros((ct, 1, 1, cm4), dtype=np.float32)
    w3[..., :cm] = rng.standard_normal((ct, 1, 1, cm), dtype=np.float32) / np.sqrt(cm)
    b3 = rng.standard_normal(ct, dtype=np.float32)
    ho, wo = abi.conv_out_size(abi.MODE_CONV, h, w, k, s, k // 2)
    res = rng.standard_normal((n, ho, wo, ct), dtype=np.float32) if use_res else None
    t = oracle.conv2d(x, wt, b1, stride=s, pad=k // 2, act1=a1)
    if cm4 != cm:
        t = np.concatenate([t, np.zeros(t.shape[:3] + (cm4 - cm,), np.float32)], axis=-1)
    want = oracle.conv2d(t, w3, b3, res=res, act2=a2)
    if cm4 == cm:  # the oracle's own fused twin
        np.testing.assert_array_equal(oracle.conv2d(x, wt, b1, stride=s, pad=k // 2, act1=a1, act2=a2, res=res, tail=(w3, b3)), want)
    got = ops.conv2d(T(x, cuda), T(wt, cuda), T(b1, cuda), stride=s, pad=k // 2, act1=a1, act2=a2,
                     res=None if res is None else T(res, cuda), tail=(T(w3, cuda), T(b3, cuda)))
    eq(got, want)


@pytest.mark.parametrize('n,h,w,c', [(2, 16, 32, 128), (1, 9, 11, 128), (2, 8, 8, 64)])
def test_attention_gate_epilogue_bit_exact(n, h, w, c, oracle, cuda):
    """x + trunk * sigmoid(conv1x1(a)) in the conv epilogue (whole 64x64 tiles take the inlined-sigmoid path, ragged ones
    the general one): both equal the oracle"""
    from aivc_amd import ops
    rng = np.random.default_rng(n * 100 + h)
    a = rng.standard_normal((n, h, w, c), dtype=np.float32)
    wt = (rng.standard_normal((c, 1, 1, c), dtype=np.float32) / np.sqrt(c)).astype(np.float32)
    b = rng.standard_normal(c, dtype=np.float32)
    trunk = rng.standard_normal((n, h, w, c), dtype=np.float32)
    x = rng.standard_normal((n, h, w, c), dtype=np.float32)
    want = oracle.conv2d(a, wt, b, act1=abi.ACT_SIGMOID, mul=trunk, res=x)
    got = ops.conv2d(T(a, cuda), T(wt, cuda), T(b, cuda), act1=abi.ACT_SIGMOID, mul=T(trunk, cuda), res=T(x, cuda))
    eq(got, want)


def test_fused_tail_is_one_launch(cuda):
    """the bottleneck-block shape takes the fused kernel (variant 190), others are declined by the library"""
    import ctypes as C
    from aivc_amd._lib import load
    x = torch.zeros((1, 16, 16, 64), device=cuda)
    w = torch.zeros((64, 3, 3, 64), device=cuda)
    b = torch.zeros(64, device=cuda)
    y = torch.zeros((1, 16, 16, 128), device=cuda)
    w3 = torch.zeros((128, 1, 1, 64), device=cuda)
    b3 = torch.zeros(128, device=cuda)
    ptr = lambda t: t.data_ptr()
    p = abi.ConvParams(abi.MODE_CONV, 3, 1, 1, 1, 16, 16, 64, 16, 16, 64, abi.ACT_LEAKY, abi.ACT_LEAKY, abi.ALGO_AUTO, 0, 0,
                       ptr(x), ptr(w), ptr(b), None, None, ptr(y), None, None, ptr(w3), ptr(b3), 128, 0)
    assert load()['aivc_conv2d_variant'](C.byref(p)) == 190
    p.tail_c_out = 64
    assert load()['aivc_conv2d_variant'](C.byref(p)) < 0
    p.tail_c_out = 128
    p.algo = abi.ALGO_DIRECT
    assert load()['aivc_conv2d_variant'](C.byref(p)) < 0


@pytest.mark.parametrize('h,w,n', [(9, 13, 2), (16, 128, 1), (35, 131, 2), (64, 64, 3)])
@pytest.mark.parametrize('use_gdn', [True, False])
def test_conv_images_bit_exact(h, w, n, use_gdn, cuda, oracle, monkeypatch):
    """aivc_conv_images (first analysis layer straight from the image sources) == oracle conv over the packed tensor
    == the HIP pack + conv pair, for 1 / 2 / 3 images, 8-bit 4:2:0 and float sources, ragged tiles"""
    import ctypes as C
    from aivc_amd import ops
    from aivc_amd._lib import load
    monkeypatch.setattr(ops, '_CONV_IMAGES_MAX', 3)  # the codec sends 3-image stacks down the pack + conv path (faster)
    rng = np.random.default_rng(h * 1000 + w + (7 if use_gdn else 0))
    hc, wc = (h + 1) // 2, (w + 1) // 2

    def planes():
        return {'y': rng.integers(0, 256, (n, h, w), dtype=np.uint8), 'u': rng.integers(0, 256, (n, hc, wc), dtype=np.uint8),
                'v': rng.integers(0, 256, (n, hc, wc), dtype=np.uint8)}
    a, b = planes(), planes()
    f = rng.standard_normal((n, h, w, 4)).astype(np.float32)
    f[..., 3] = 0.0
    dev = lambda p: {k: torch.from_numpy(p[k]).to(cuda) for k in 'yuv'}
    for parts_np in ([a], [f], [a, b], [a, f], [a, b, a], [f, a, None]):
        ni = len(parts_np)
        wt = np.zeros((64, 5, 5, 4 * ni), np.float32)
        for i in range(ni):
            wt[..., 4 * i:4 * i + 3] = rng.standard_normal((64, 5, 5, 3)).astype(np.float32) / np.sqrt(75 * ni)
        bias = rng.standard_normal(64, dtype=np.float32)
        g = None
        if use_gdn:
            g = ((np.abs(rng.standard_normal(64)) + 0.2).astype(np.float32),
                 (np.abs(rng.standard_normal((64, 64))) * 0.05).astype(np.float32), False)
        act1 = 0 if use_gdn else abi.ACT_LEAKY
        packed = oracle.pack_images(parts_np, h, w)
        want = oracle.conv2d(packed, wt, bias, stride=2, pad=2, act1=act1, gdn=g)
        np.testing.assert_array_equal(oracle.conv_images(parts_np, h, w, wt, bias, act1=act1, gdn=g), want)
        parts_t = [dev(p) if isinstance(p, dict) else (None if p is None else torch.from_numpy(p).to(cuda)) for p in parts_np]
        gt = None if g is None else (T(g[0], cuda), T(g[1], cuda), False)
        before = load()['aivc_abi_version']()
        stack = ops.ImageStack(parts_t, h, w, cuda)
        got = ops.conv2d(stack, T(wt, cuda), T(bias, cuda), stride=2, pad=2, act1=act1, gdn=gt)
        assert stack._packed is None, 'the fused kernel must have taken this layer (no packed tensor)'
        eq(got, want)
        # the two-call path it replaces
        two = ops.conv2d(ops.pack_images(parts_t, h, w, cuda), T(wt, cuda), T(bias, cuda), stride=2, pad=2, act1=act1, gdn=gt)
        eq(two, want)
        assert before == abi.ABI_VERSION


def test_conv_images_declines_other_layers(cuda):
    """layers outside the kernel's coverage fall back to pack + conv (same result path as before)"""
    from aivc_amd import ops
    rng = np.random.default_rng(5)
    f = torch.from_numpy(rng.standard_normal((1, 12, 12, 4)).astype(np.float32)).to(cuda)
    stack = ops.ImageStack([f], 12, 12, cuda)
    w = torch.from_numpy(rng.standard_normal((32, 3, 3, 4)).astype(np.float32)).to(cuda)
    y = ops.conv2d(stack, w, None, stride=1, pad=1)
    assert stack._packed is not None and tuple(y.shape) == (1, 12, 12, 32)


@pytest.mark.parametrize('h,w', [(9, 13), (16, 32), (35, 1030)])
def test_pack_images_bit_exact(h, w, cuda, oracle):
    """aivc_pack_images (padded multi-image input of the first convs) == oracle twin == per-image conversion"""
    from aivc_amd import ops
    rng = np.random.default_rng(h * 1000 + w)
    hc, wc = (h + 1) // 2, (w + 1) // 2
    n = 2

    def planes():
        return {'y': rng.integers(0, 256, (n, h, w), dtype=np.uint8), 'u': rng.integers(0, 256, (n, hc, wc), dtype=np.uint8),
                'v': rng.integers(0, 256, (n, hc, wc), dtype=np.uint8)}
    a, b = planes(), planes()
    f = rng.standard_normal((n, h, w, 4)).astype(np.float32)
    dev = lambda p: {k: torch.from_numpy(p[k]).to(cuda) for k in 'yuv'}
    for parts_np in ([a], [a, None], [a, b, None], [a, f], [a, b, a], [None, f, b]):
        parts_t = [dev(p) if isinstance(p, dict) else (None if p is None else torch.from_numpy(p).to(cuda)) for p in parts_np]
        got = ops.pack_images(parts_t, h, w, cuda)
        want = oracle.pack_images(parts_np, h, w)
        assert got._aivc_cmap == tuple(4 * i + c for i in range(len(parts_np)) for c in range(3))
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        # and the same as the per-image kernels
        for i, p in enumerate(parts_np):
            if isinstance(p, dict):
                ref = ops.yuv420_to_444(parts_t[i]['y'], parts_t[i]['u'], parts_t[i]['v'], c_store=4)
                assert torch.equal(got[..., 4 * i:4 * i + 4], ref)
            elif p is None:
                assert not got[..., 4 * i:4 * i + 4].any()
            else:
                assert torch.equal(got[..., 4 * i:4 * i + 3], parts_t[i][..., :3]) and not got[..., 4 * i + 3].any()


def test_frame_batch_entropy_kernels_equal_per_frame_calls(oracle, cuda):
    """aivc_laplace_cdf_windows_batch / laplace_bounds_batch / table_bounds_batch / scatter_symbols_batch (one launch per
    frame batch, per-frame map lists in a device table) == the single-frame entry points frame by frame, incl. frames
    with no coded map, and == the oracle's twins"""
    from aivc_amd import ops
    rng = np.random.default_rng(21)
    n, h, w, c = 5, 7, 9, 16
    npix = h * w
    sig = (np.abs(rng.standard_normal((n, h, w, c))) * 2 + 0.05).astype(np.float32)
    q = np.clip(np.rint(rng.standard_normal((n, h, w, c)) * sig), -256, 256).astype(np.int16)
    maps = [[0, 3, 15], [], [1], list(range(c)), [2, 14]]
    sd, qd = T(sig, cuda), T(q, cuda)
    # bounds
    allb, offs = ops.laplace_bounds_batch(sd, qd, maps)
    for f, m in enumerate(maps):
        if m:
            eq(allb[offs[f]:offs[f] + len(m) * npix], ops.laplace_bounds(sd[f:f + 1], qd[f:f + 1], m))
            np.testing.assert_array_equal(allb[offs[f]:offs[f] + len(m) * npix].cpu().numpy().view(np.uint32),
                                          oracle.laplace_bounds(sig[f:f + 1], q[f:f + 1], m))
    # windows + sigma per position
    total = sum(len(m) for m in maps) * npix
    win = torch.zeros((total, abi.CDF_WIN), dtype=torch.int16, device=cuda)
    sp = torch.zeros(total, dtype=torch.float32, device=cuda)
    offs2, tab = ops.laplace_cdf_windows_batch(sd, maps, (win, sp))
    assert offs2 == offs
    for f, m in enumerate(maps):
        if m:
            w1, s1 = ops.laplace_cdf_windows(sd[f:f + 1], m)
            eq(win[offs[f]:offs[f] + len(m) * npix], w1)
            eq(sp[offs[f]:offs[f] + len(m) * npix], s1)
    # pmf bounds of every channel
    table = T(rng.integers(0, 65535, (c, abi.CDF_ROW)).astype(np.uint16).view(np.int16), cuda)
    zb = ops.table_bounds_batch(table, qd)
    for f in range(n):
        eq(zb[f], ops.table_bounds(table, qd[f:f + 1]))
    # scatter: symbols in stream order -> [n, npix, c]
    sym = torch.cat([(qd[f].reshape(npix, c)[:, m].T.reshape(-1).to(torch.int32) + 256).to(torch.int16) for f, m in enumerate(maps) if m])
    back = ops.scatter_symbols_batch(sym, maps, n, npix, c, table=tab).view(n, h, w, c)
    want = np.zeros_like(q)
    for f, m in enumerate(maps):
        want[f][..., m] = q[f][..., m]
    eq(back, want)
