"""CLIC-2021 metrics on in-memory planes -- mirror of src/clic21/metrics.py (evaluate / mse / mse2psnr / msssim),
the arithmetic behind evaluate.py's four numbers, without the PNG round trip: `evaluate` takes
{name: plane} dicts of uint8 / float planes (numpy or CUDA tensors, values 0..255) instead of PNG paths.
(The reference converts each 8-bit plane to 3 identical RGB channels first, which changes neither the PSNR
nor the size-weighted MS-SSIM.)"""
import json

import numpy as np
import torch

from .. import ops
from .msssim import MultiScaleSSIM


def _plane64(p):
    t = p if torch.is_tensor(p) else torch.from_numpy(np.ascontiguousarray(p))
    t = t if t.is_cuda else t.to('cuda')
    return t.reshape(t.shape[-2], t.shape[-1]).to(torch.float64).contiguous()


def mse(image0, image1):
    """sum of squared differences (metrics.py:58-59; the name is the reference's)"""
    return float(ops.sq_err(_plane64(image1), _plane64(image0)).item())


def mse2psnr(mse_value):
    return 20. * np.log10(255.) - 10. * np.log10(mse_value)


def msssim(image0, image1):
    a, b = _plane64(image0), _plane64(image1)
    return MultiScaleSSIM(a[None, :, :, None], b[None, :, :, None])


def evaluate(submission_images, target_images, settings={}, logger=None):
    """metrics.py:6-56 with planes instead of PNG paths -> {'PSNR', 'MSSSIM', 'MSSSIM_dB'}"""
    if settings is None:
        settings = {}
    if isinstance(settings, str):
        try:
            settings = json.loads(settings)
        except json.JSONDecodeError:
            settings = {}
    metrics = settings.get('metrics', ['PSNR', 'MSSSIM'])
    num_dims = 0
    sqerror_values, msssim_values = [], []
    for name in target_images:
        image0, image1 = target_images[name], submission_images[name]
        size = int(np.prod(tuple(image0.shape)))
        num_dims += size
        if 'PSNR' in metrics:
            sqerror_values.append(mse(image1, image0))
        if 'MSSSIM' in metrics:
            value = msssim(image0, image1) * size
            if np.isnan(value):
                value = 0.0
                if logger:
                    logger.warning('Evaluation of MSSSIM for `%s` returned NaN. Assuming MSSSIM is zero.' % name)
            msssim_values.append(value)
    results = {}
    if 'PSNR' in metrics:
        results['PSNR'] = mse2psnr(np.sum(sqerror_values) / num_dims)
    if 'MSSSIM' in metrics:
        results['MSSSIM'] = np.sum(msssim_values) / num_dims
        results['MSSSIM_dB'] = -10 * np.log10(1 - results.get('MSSSIM'))
    return results
