"""CLIC-2021 MS-SSIM scorer on the device -- mirror of src/clic21/msssim.py:28-178 (MultiScaleSSIM and its
helpers), the scorer behind evaluate.py's 'MS-SSIM' lines.  fp64 like the original (numpy + fftconvolve):
parity 1e-9 on the final score against the reference's own outputs (tests/golden/metrics.npz)."""
import numpy as np
import torch

from .. import ops

_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _fspecial_gauss_1d(size, sigma):
    """separable factor of _FSpecialGauss (msssim.py:28-40): exp(-x^2 / 2 sigma^2) on the same grid, normalised"""
    radius = size // 2
    offset, start, stop = 0.0, -radius, radius + 1
    if size % 2 == 0:
        offset = 0.5
        stop -= 1
    x = np.arange(offset + start, stop, 1.0)
    g = np.exp(-(x ** 2) / (2.0 * sigma ** 2))
    return g / g.sum()


def _to_planes(img):
    """[batch, h, w, depth] array / tensor -> float64 CUDA planes [batch*depth, h, w]"""
    t = img if torch.is_tensor(img) else torch.from_numpy(np.ascontiguousarray(img))
    if t.dim() != 4:
        raise RuntimeError('Input images must have four dimensions, not %d' % t.dim())
    t = t.to('cuda' if not t.is_cuda else t.device)
    b, h, w, d = t.shape
    return t.permute(0, 3, 1, 2).reshape(b * d, h, w).to(torch.float64).contiguous()


def _SSIMForMultiScale(img1, img2, max_val=255, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """msssim.py:43-113 on planes: -> (mean SSIM, mean contrast) as float64 0-d CUDA tensors"""
    p1 = img1 if (torch.is_tensor(img1) and img1.dim() == 3) else _to_planes(img1)
    p2 = img2 if (torch.is_tensor(img2) and img2.dim() == 3) else _to_planes(img2)
    if p1.shape != p2.shape:
        raise RuntimeError('Input images must have the same shape (%s vs. %s).' % (tuple(p1.shape), tuple(p2.shape)))
    _, height, width = p1.shape
    size = min(filter_size, height, width)
    sigma = size * filter_sigma / filter_size if filter_size else 0
    win = _fspecial_gauss_1d(size, sigma) if filter_size else np.ones(1)
    m = ops.ssim_means(p1, p2, win, (k1 * max_val) ** 2, (k2 * max_val) ** 2)
    return m[:, 0].mean(), m[:, 1].mean()


def MultiScaleSSIM(img1, img2, max_val=255, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, weights=None):
    """msssim.py:116-178 -> Python float"""
    p1, p2 = _to_planes(img1), _to_planes(img2)
    if p1.shape != p2.shape:
        raise RuntimeError('Input images must have the same shape (%s vs. %s).' % (tuple(p1.shape), tuple(p2.shape)))
    weights = np.array(weights if weights else _WEIGHTS)
    levels = weights.size
    mssim, mcs = [], []
    for _ in range(levels):
        s, c = _SSIMForMultiScale(p1, p2, max_val=max_val, filter_size=filter_size, filter_sigma=filter_sigma, k1=k1, k2=k2)
        mssim.append(s)
        mcs.append(c)
        p1, p2 = ops.pool2x2(p1, 1), ops.pool2x2(p2, 1)
    vals = torch.stack(mssim + mcs).cpu().numpy()  # one sync
    mssim, mcs = vals[:levels], vals[levels:]
    with np.errstate(invalid='ignore'):  # a negative contrast term gives NaN, as in the reference (metrics.py:40-45 zeroes it)
        return float(np.prod(mcs[0:levels - 1] ** weights[0:levels - 1]) * (mssim[levels - 1] ** weights[levels - 1]))
