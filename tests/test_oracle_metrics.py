"""The CPU restatement of the two MS-SSIM variants (oracle/metrics.py + the C twins) against outputs of the
reference's own functions (tests/golden/metrics.npz, tools/gen_golden_metrics.py)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metrics.npz'))
CASES = list(range(int(G['n_cases'])))


@pytest.mark.parametrize('i', CASES)
def test_clic_variant_matches_reference_fp64(i):
    from oracle import metrics, oracle
    a, b = G['a%d' % i].astype(np.float64)[None], G['b%d' % i].astype(np.float64)[None]
    assert oracle.sq_err(a, b)[0] == float(G['clic_sqerr%d' % i])  # integers: exact
    m = oracle.ssim_means(a, b, metrics.window_clic(11, 1.5), (0.01 * 255) ** 2, (0.03 * 255) ** 2)[0]
    np.testing.assert_allclose(m, G['clic_ssim0_%d' % i], rtol=0, atol=1e-12)
    # the reference convolves with FFTs: agreement is at the 1e-12 level, not bitwise
    assert abs(metrics.msssim_clic(a, b) - float(G['clic_msssim%d' % i])) < 1e-11


@pytest.mark.parametrize('i', CASES)
def test_torch_variant_matches_reference_fp32(i):
    from oracle import metrics, oracle
    a, b = (G['a%d' % i].astype(np.float32) / 255.0)[None], (G['b%d' % i].astype(np.float32) / 255.0)[None]
    m = oracle.ssim_means(a, b, metrics.window_torch(11), 0.01 ** 2, 0.03 ** 2)[0]
    # the reference works in fp32 (ATen conv): its own rounding is the tolerance
    np.testing.assert_allclose(m, G['torch_ssim0_%d' % i], rtol=0, atol=2e-5)
    assert abs(metrics.msssim_torch(a, b, 1.0) - float(G['torch_msssim%d' % i])) < 2e-5


def test_pooling_edge_rules():
    from oracle import oracle
    x = np.arange(15, dtype=np.float64).reshape(1, 3, 5)
    p0, p1 = oracle.pool2x2(x, 0)[0], oracle.pool2x2(x, 1)[0]
    assert p0.shape == p1.shape == (2, 3)
    assert p0[0, 0] == p1[0, 0] == (0 + 1 + 5 + 6) / 4
    # last column / row: mirror without the border sample (index n-2) vs repeat the border (index n-1)
    assert p0[0, 2] == (4 + 3 + 9 + 8) / 4 and p1[0, 2] == (4 + 4 + 9 + 9) / 4
    assert p0[1, 0] == (10 + 11 + 5 + 6) / 4 and p1[1, 0] == (10 + 11 + 10 + 11) / 4


def test_evaluate_arithmetic_matches_reference():
    from oracle import metrics, oracle
    num, sq, ms = 0, 0.0, 0.0
    for i in CASES:
        a, b = G['a%d' % i].astype(np.float64)[None], G['b%d' % i].astype(np.float64)[None]
        num += a.size
        sq += oracle.sq_err(a, b)[0]
        ms += metrics.msssim_clic(a, b) * a.size
    assert abs(20 * np.log10(255.) - 10 * np.log10(sq / num) - float(G['eval_psnr'])) < 1e-10
    assert abs(ms / num - float(G['eval_msssim'])) < 1e-11
