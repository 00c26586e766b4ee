"""True parity of the range coder against the real `torchac` package (fab-jul/torchac) -- runs wherever the wheel
is importable, skipped elsewhere (it is absent from the build image and the GPU box: SURVEY.md 8c).  The reference's
call sites: src/real_life/bitstream.py:281 (encode_float_cdf(cdf, sym, check_input_bounds=True)), :454-456
(decode_float_cdf(cdf, bytes, needs_normalization=True)), :482 (decode_float_cdf(cdf, bytes))."""
import numpy as np
import pytest

torchac = pytest.importorskip('torchac')
import torch  # noqa: E402


def _float_cdf(sigma):
    """what ArithmeticCoder.get_y_cdf hands to torchac: torch's Laplace.cdf at k - 256.5, [N, 514] fp32"""
    b = torch.from_numpy(sigma).reshape(-1, 1) / torch.sqrt(torch.tensor([2.0]))
    t = torch.arange(514, dtype=torch.float32) - 256.5
    return torch.distributions.Laplace(torch.zeros_like(b), b).cdf(t[None, :])


def _int_rows(cdf_float):
    """torchac's float -> int16 normalisation (PRECISION 16), as uint16 rows of AIVC_CDF_ROW"""
    from aivc_amd import abi
    lp = cdf_float.shape[-1]
    c = cdf_float.mul(2 ** 16 - (lp - 1)).round().to(torch.int16) + torch.arange(lp, dtype=torch.int16)
    rows = np.zeros((c.shape[0], abi.CDF_ROW), np.uint16)
    rows[:, :lp] = c.numpy().view(np.uint16)
    return rows


@pytest.mark.parametrize('sigma', [1e-4, 0.05, 0.7, 3.0, 30.0, 148.0])
def test_oracle_coder_equals_torchac(sigma, oracle):
    rng = np.random.default_rng(int(sigma * 1000) + 1)
    n = 4000
    sig = np.full(n, sigma, np.float32) * np.exp(rng.uniform(-0.5, 0.5, n)).astype(np.float32)
    q = np.clip(np.rint(rng.laplace(0, 1, n) * sig / np.sqrt(2)), -256, 256).astype(np.int16)
    q[:10] = [-256, -255, -33, -32, 30, 31, 32, 254, 255, 256]  # alphabet / window edges incl. max_symbol 512
    sym = (q.astype(np.int32) + 256).astype(np.int16)
    cdf = _float_cdf(sig)
    ref_bytes = torchac.encode_float_cdf(cdf, torch.from_numpy(sym), check_input_bounds=True)
    rows = _int_rows(cdf)
    ar = np.arange(n)
    s = sym.astype(np.int64)
    hi = np.where(s == 512, 0, rows[ar, np.minimum(s + 1, 513)]).astype(np.uint32)
    bounds = rows[ar, s].astype(np.uint32) | (hi << 16)
    assert oracle.range_encode(bounds) == ref_bytes
    # both decoders decode both streams
    np.testing.assert_array_equal(oracle.range_decode(ref_bytes, rows, n).astype(np.int16), sym)
    back = torchac.decode_float_cdf(cdf, oracle.range_encode(bounds), needs_normalization=True)
    np.testing.assert_array_equal(back.numpy().reshape(-1), sym)


@pytest.mark.gpu
def test_hip_coder_equals_torchac(cuda):
    from aivc_amd import ops
    rng = np.random.default_rng(9)
    n = 3000
    sig = np.exp(rng.uniform(np.log(0.05), np.log(40.0), n)).astype(np.float32)
    q = np.clip(np.rint(rng.laplace(0, 1, n) * sig / np.sqrt(2)), -256, 256).astype(np.int16)
    q[:4] = [-256, 255, 256, 0]
    sym = (q.astype(np.int32) + 256).astype(np.int16)
    cdf = _float_cdf(sig)
    ref_bytes = torchac.encode_float_cdf(cdf, torch.from_numpy(sym), check_input_bounds=True)
    rows = _int_rows(cdf)
    got = ops.range_decode([ref_bytes], torch.from_numpy(rows.view(np.int16)).to(cuda), [0], [n], [0])[0]
    np.testing.assert_array_equal(got.cpu().numpy(), sym)
