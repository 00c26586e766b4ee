"""MS-SSIM on the device -- mirror of src/func_util/ms_ssim.py (same names, arguments and defaults; the
torch re-implementation of pytorch-msssim the reference uses for its per-frame 'ms_ssim' figure,
src/model_mngt/loss_function.py:437-470).

The five Gaussian-window correlations per scale, the SSIM / contrast maps and their means run in ONE HIP
kernel per scale (aivc_ssim_means, fp64, fixed-order reductions), the 2x2 pooling between scales in another
(aivc_pool2x2); nothing is materialised but the pyramid itself.  The reference computes in fp32 with ATen
convolutions whose summation order is unspecified: parity is |delta| <= 2e-5 on the final score
(tests/test_gpu_metrics.py), against fixtures produced by the reference itself (tests/golden/metrics.npz).
"""
from math import exp

import numpy as np
import torch

from .. import ops

_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gaussian(window_size, sigma):
    """1-D window as the reference builds it (ms_ssim.py:24-27): fp32 tensor, normalised in fp32"""
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel=1):
    """[channel,1,ws,ws] fp32 window (ms_ssim.py:30-35); the kernels use its separable form"""
    _1d = gaussian(window_size, 1.5).unsqueeze(1)
    _2d = _1d.mm(_1d.t()).float().unsqueeze(0).unsqueeze(0)
    return _2d.expand(channel, 1, window_size, window_size).contiguous()


def _planes(img):
    """[b,c,h,w] tensor -> float64 CUDA planes [b*c, h, w]"""
    if img.dim() != 4:
        raise ValueError('expected a [b,c,h,w] tensor')
    b, c, h, w = img.shape
    return img.reshape(b * c, h, w).to(torch.float64).contiguous(), b, c


def _dynamic_range(img1, val_range):
    if val_range is not None:
        return val_range
    max_val = 255 if torch.max(img1) > 128 else 1
    min_val = -1 if torch.min(img1) < -0.5 else 0
    return max_val - min_val


def _ssim_planes(p1, p2, b, c, window_size, L, size_average):
    _, h, w = p1.shape
    real_size = min(window_size, h, w)
    win = gaussian(real_size, 1.5).numpy().astype(np.float64)
    m = ops.ssim_means(p1, p2, win, (0.01 * L) ** 2, (0.03 * L) ** 2)  # [b*c, 2]
    cs = m[:, 1].mean()  # the reference's cs is always the global mean (ms_ssim.py:79)
    ret = m[:, 0].mean() if size_average else m[:, 0].reshape(b, c).mean(dim=1)
    return ret, cs


def ssim(img1, img2, window_size=11, window=None, size_average=True, full=False, val_range=None):
    """ms_ssim.py:37-90.  `window` is accepted for signature compatibility; only its size is used (the
    reference's windows are always create_window(size))."""
    L = _dynamic_range(img1, val_range)
    p1, b, c = _planes(img1)
    p2, _, _ = _planes(img2)
    if window is not None:
        window_size = window.shape[-1]
    ret, cs = _ssim_planes(p1, p2, b, c, window_size, L, size_average)
    ret, cs = ret.to(torch.float32), cs.to(torch.float32)
    return (ret, cs) if full else ret


def msssim(img1, img2, window_size=11, size_average=True, val_range=None, normalize=False, full=True):
    """ms_ssim.py:93-150: five scales, reflection-padded 2x2 mean between them, prod(mcs[:-1]**w[:-1]) * mssim[-1]**w[-1]"""
    p1, b, c = _planes(img1)
    p2, _, _ = _planes(img2)
    weights = torch.tensor(_WEIGHTS, dtype=torch.float64, device=p1.device)
    mssim, mcs = [], []
    for _ in range(len(_WEIGHTS)):
        # the reference re-derives the range from the CURRENT (pooled) image when val_range is None
        L = val_range if val_range is not None else _dynamic_range(p1, None)
        sim, cs = _ssim_planes(p1, p2, b, c, window_size, L, size_average)
        mssim.append(sim)
        mcs.append(cs if size_average else cs.expand_as(sim))
        p1, p2 = ops.pool2x2(p1, 0), ops.pool2x2(p2, 0)
    mssim, mcs = torch.stack(mssim), torch.stack(mcs)
    if normalize:
        mssim, mcs = (mssim + 1) / 2, (mcs + 1) / 2
    w = weights if size_average else weights[:, None]
    pow1, pow2 = mcs ** w, mssim ** w
    return (torch.prod(pow1[:-1], dim=0) * pow2[-1]).to(torch.float32)


class MSSSIM(torch.nn.Module):
    """ms_ssim.py:181-194"""

    def __init__(self, window_size=11, size_average=True, channel=3, max_val=None):
        super(MSSSIM, self).__init__()
        self.window_size = window_size
        self.size_average = size_average
        self.channel = channel
        self.val_range = max_val

    def forward(self, img1, img2):
        return msssim(img1, img2, window_size=self.window_size, size_average=self.size_average,
                      val_range=self.val_range, full=True)
