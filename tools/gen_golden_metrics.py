#!/usr/bin/env python3
"""Generate tests/golden/metrics.npz by IMPORTING the reference's two MS-SSIM implementations
(src/func_util/ms_ssim.py, torch fp32; src/clic21/msssim.py + metrics.py, numpy fp64) and running them on
seeded 8-bit plane pairs.  Build container only (needs /root/reference); the fixture holds inputs and the
reference's outputs, nothing of its source.

    python tools/gen_golden_metrics.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = '/root/reference/src'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'metrics.npz')


def plane_pair(h, w, seed, noise):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = 128 + 60 * np.sin(2 * np.pi * x / 37.0) + 45 * np.cos(2 * np.pi * y / 23.0) + rng.normal(0, 6, (h, w))
    a = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    b = np.clip(np.rint(base + rng.normal(0, noise, (h, w))), 0, 255).astype(np.uint8)
    return a, b


def main():
    sys.path.insert(0, REF)
    from func_util import ms_ssim as ref_t
    from clic21 import msssim as ref_c
    from clic21 import metrics as ref_m
    out = {}
    cases = [(65, 97, 1, 8.0), (176, 144, 2, 3.0), (40, 56, 3, 12.0), (64, 48, 4, 1.0)]
    out['n_cases'] = len(cases)
    for i, (h, w, seed, noise) in enumerate(cases):
        a, b = plane_pair(h, w, seed, noise)
        out['a%d' % i], out['b%d' % i] = a, b
        # torch variant: [1,1,h,w] float32 in [0,1], val_range=1 (as MSSSIMLoss uses it, loss_function.py:443)
        ta = torch.from_numpy(a.astype(np.float32) / 255.0)[None, None]
        tb = torch.from_numpy(b.astype(np.float32) / 255.0)[None, None]
        out['torch_msssim%d' % i] = ref_t.msssim(ta, tb, val_range=1.0).numpy()
        s, c = ref_t.ssim(ta, tb, full=True, val_range=1.0)
        out['torch_ssim0_%d' % i] = np.array([s.item(), c.item()])
        # CLIC variant: [1,h,w,1] float 0..255
        ca, cb = a.astype(np.float32)[None, :, :, None], b.astype(np.float32)[None, :, :, None]
        out['clic_msssim%d' % i] = np.float64(ref_c.MultiScaleSSIM(ca, cb))
        out['clic_ssim0_%d' % i] = np.array(ref_c._SSIMForMultiScale(ca, cb), dtype=np.float64)
        out['clic_sqerr%d' % i] = np.float64(ref_m.mse(ca, cb))
    # the three aggregate numbers of metrics.evaluate over all cases treated as one set of images
    # (evaluate() itself opens PNG files; this is its arithmetic, metrics.py:36-55, on the same arrays)
    num_dims = sum(int(out['a%d' % i].size) for i in range(len(cases)))
    sq = sum(float(out['clic_sqerr%d' % i]) for i in range(len(cases)))
    ms = sum(float(out['clic_msssim%d' % i]) * out['a%d' % i].size for i in range(len(cases)))
    out['eval_psnr'] = np.float64(ref_m.mse2psnr(sq / num_dims))
    out['eval_msssim'] = np.float64(ms / num_dims)
    out['eval_msssim_db'] = np.float64(-10 * np.log10(1 - ms / num_dims))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')
    for k in sorted(out):
        if not k[0] in 'ab':
            print(k, out[k])


if __name__ == '__main__':
    main()
