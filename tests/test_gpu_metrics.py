"""On-device quality metrics (aivc_ssim_means / aivc_pool2x2 / aivc_sq_err and the two MS-SSIM front ends built
on them) against the CPU oracle and against outputs of the reference's own functions (tests/golden/metrics.npz).
fp64 kernels: tolerance 1e-12 against the fp64 oracle (separable vs 2-D window summation order), 2e-5 against
the reference's fp32 torch variant (its own rounding), 1e-9 against its fp64 CLIC variant (FFT convolution)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metrics.npz'))
CASES = list(range(int(G['n_cases'])))


@pytest.fixture(scope='module')
def cuda():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


@pytest.mark.parametrize('h,w,ws', [(16, 16, 11), (11, 11, 11), (37, 53, 11), (9, 30, 9), (5, 7, 5), (3, 4, 3), (64, 129, 6), (2, 2, 1)])
def test_ssim_scale_pool_and_sqerr_match_oracle(h, w, ws, cuda):
    from aivc_amd import ops
    from oracle import metrics, oracle
    rng = np.random.default_rng(h * 100 + w)
    a = rng.uniform(0, 255, (3, h, w))
    b = np.clip(a + rng.normal(0, 9, a.shape), 0, 255)
    win = metrics.window_clic(ws, ws * 1.5 / 11)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    ta, tb = torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)
    got = ops.ssim_means(ta, tb, win, c1, c2).cpu().numpy()
    np.testing.assert_allclose(got, oracle.ssim_means(a, b, win, c1, c2), rtol=0, atol=1e-12)
    for edge in (0, 1):
        np.testing.assert_array_equal(ops.pool2x2(ta, edge).cpu().numpy(), oracle.pool2x2(a, edge))
    se, se_ref = ops.sq_err(ta, tb).item(), oracle.sq_err(a, b)[0]
    assert abs(se - se_ref) <= 1e-12 * se_ref
    # integer-valued planes: the sum of squares is exact in fp64
    ia, ib = np.rint(a), np.rint(b)
    assert ops.sq_err(torch.from_numpy(ia).to(cuda), torch.from_numpy(ib).to(cuda)).item() == oracle.sq_err(ia, ib)[0]


@pytest.mark.parametrize('i', CASES)
def test_both_msssim_front_ends_match_reference_outputs(i, cuda):
    from aivc_amd.clic21 import metrics as clic_metrics
    from aivc_amd.clic21 import msssim as clic
    from aivc_amd.func_util import ms_ssim
    from oracle import metrics as ometrics
    a8, b8 = G['a%d' % i], G['b%d' % i]
    # CLIC scorer (fp64)
    ca, cb = a8.astype(np.float32)[None, :, :, None], b8.astype(np.float32)[None, :, :, None]
    got = clic.MultiScaleSSIM(ca, cb)
    assert abs(got - float(G['clic_msssim%d' % i])) < 1e-9
    assert abs(got - ometrics.msssim_clic(a8[None].astype(np.float64), b8[None].astype(np.float64))) < 1e-12
    s, c = clic._SSIMForMultiScale(ca, cb)
    np.testing.assert_allclose([s.item(), c.item()], G['clic_ssim0_%d' % i], rtol=0, atol=1e-11)
    assert clic_metrics.mse(a8, b8) == float(G['clic_sqerr%d' % i])
    # torch variant (fp32 reference)
    ta = torch.from_numpy(a8.astype(np.float32) / 255.0)[None, None].to(cuda)
    tb = torch.from_numpy(b8.astype(np.float32) / 255.0)[None, None].to(cuda)
    assert abs(ms_ssim.msssim(ta, tb, val_range=1.0).item() - float(G['torch_msssim%d' % i])) < 2e-5
    s, c = ms_ssim.ssim(ta, tb, full=True, val_range=1.0)
    np.testing.assert_allclose([s.item(), c.item()], G['torch_ssim0_%d' % i], rtol=0, atol=2e-5)
    assert abs(ms_ssim.MSSSIM(max_val=1.)(ta, tb).item() - float(G['torch_msssim%d' % i])) < 2e-5


def test_evaluate_on_planes_gives_the_reference_numbers(cuda):
    from aivc_amd.clic21.metrics import evaluate
    target = {str(i): G['a%d' % i] for i in CASES}
    submit = {str(i): torch.from_numpy(G['b%d' % i]).to(cuda) for i in CASES}
    r = evaluate(submit, target)
    assert abs(r['PSNR'] - float(G['eval_psnr'])) < 1e-9
    assert abs(r['MSSSIM'] - float(G['eval_msssim'])) < 1e-9
    assert abs(r['MSSSIM_dB'] - float(G['eval_msssim_db'])) < 1e-6


def test_full_size_plane_properties(cuda):
    """1080p: identical planes score exactly 1, the score falls monotonically with the noise level"""
    from aivc_amd.clic21 import msssim as clic
    g = torch.Generator(device=cuda).manual_seed(5)
    a = torch.randint(0, 256, (1, 1080, 1920, 1), generator=g, device=cuda, dtype=torch.int32).to(torch.float32)
    assert clic.MultiScaleSSIM(a, a) == 1.0
    prev = 1.0
    for sd in (2.0, 8.0, 32.0):
        b = (a + torch.randn(a.shape, generator=g, device=cuda) * sd).clamp(0, 255).round()
        cur = clic.MultiScaleSSIM(a, b)
        assert cur < prev
        prev = cur
