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
    """metrics.py:6-56 with planes instead of PNG paths -> {'PSNR', 'MSSSIM', 'MSSSIM_dB'}.
    Squared errors and MS-SSIM scores are pooled over all planes weighted by their sample counts."""
    if isinstance(settings, str):
        try:
            settings = json.loads(settings)
        except json.JSONDecodeError:
            settings = {}
    wanted = (settings or {}).get('metrics', ['PSNR', 'MSSSIM'])
    want_psnr, want_ms = 'PSNR' in wanted, 'MSSSIM' in wanted
    samples, sq_total, ms_total = 0, 0.0, 0.0
    for name, ref_plane in target_images.items():
        got_plane = submission_images[name]
        count = int(np.prod(tuple(ref_plane.shape)))
        samples += count
        if want_psnr:
            sq_total += mse(got_plane, ref_plane)
        if want_ms:
            score = msssim(ref_plane, got_plane)
            if np.isnan(score):  # a negative contrast term at a coarse scale; the reference counts it as zero
                score = 0.0
                if logger:
                    logger.warning('Evaluation of MSSSIM for `%s` returned NaN. Assuming MSSSIM is zero.' % name)
            ms_total += score * count
    results = {}
    if want_psnr:
        results['PSNR'] = mse2psnr(sq_total / samples)
    if want_ms:
        results['MSSSIM'] = ms_total / samples
        results['MSSSIM_dB'] = -10 * np.log10(1 - results['MSSSIM'])
    return results
