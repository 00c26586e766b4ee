"""CPU restatement of the two MS-SSIM variants of the reference for the parity tests -- TEST INFRASTRUCTURE,
never imported by the product.  Follows src/func_util/ms_ssim.py:37-150 (variant 'torch': fixed sigma 1.5 window
centred on ws // 2, ReflectionPad2d before the 2x2 mean, result cast to fp32) and src/clic21/msssim.py:28-178
(variant 'clic': fspecial window with scaled sigma, scipy 'reflect' edge rule) on top of the C twins
aivc_ssim_means_ref / aivc_pool2x2_ref / aivc_sq_err_ref (literal 2-D window loops, fp64).
Pinned by tests/golden/metrics.npz = outputs of the reference's own functions (tools/gen_golden_metrics.py)."""
from math import exp

import numpy as np

from . import oracle

WEIGHTS = np.array([0.0448, 0.2856, 0.3001, 0.2363, 0.1333])


def window_torch(ws):
    g = np.array([exp(-(x - ws // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(ws)], np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float64)


def window_clic(size, sigma):
    radius = size // 2
    offset, start, stop = 0.0, -radius, radius + 1
    if size % 2 == 0:
        offset = 0.5
        stop -= 1
    x = np.arange(offset + start, stop, 1.0)
    g = np.exp(-(x ** 2) / (2.0 * sigma ** 2))
    return g / g.sum()


def msssim_torch(a, b, val_range=1.0):
    """a, b: [n,h,w] arrays -> scalar (size_average=True)"""
    p1, p2 = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ms, mc = [], []
    for _ in range(5):
        ws = min(11, p1.shape[1], p1.shape[2])
        m = oracle.ssim_means(p1, p2, window_torch(ws), (0.01 * val_range) ** 2, (0.03 * val_range) ** 2)
        ms.append(m[:, 0].mean())
        mc.append(m[:, 1].mean())
        p1, p2 = oracle.pool2x2(p1, 0), oracle.pool2x2(p2, 0)
    ms, mc = np.array(ms), np.array(mc)
    with np.errstate(invalid='ignore'):
        return float(np.prod(mc[:-1] ** WEIGHTS[:-1]) * ms[-1] ** WEIGHTS[-1])


def msssim_clic(a, b, max_val=255.0):
    p1, p2 = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ms, mc = [], []
    for _ in range(5):
        size = min(11, p1.shape[1], p1.shape[2])
        m = oracle.ssim_means(p1, p2, window_clic(size, size * 1.5 / 11), (0.01 * max_val) ** 2, (0.03 * max_val) ** 2)
        ms.append(m[:, 0].mean())
        mc.append(m[:, 1].mean())
        p1, p2 = oracle.pool2x2(p1, 1), oracle.pool2x2(p2, 1)
    ms, mc = np.array(ms), np.array(mc)
    with np.errstate(invalid='ignore'):
        return float(np.prod(mc[:-1] ** WEIGHTS[:-1]) * ms[-1] ** WEIGHTS[-1])
