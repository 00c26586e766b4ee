#!/usr/bin/env python3
"""A/B of the two versions of the fp32 contract on the stride-1 3x3 layers of the 1080p workload (run on the GPU box):
time per launch, TFLOP/s in direct-equivalent FLOPs (2 * 9 * c_in * c_out per output pixel) for both, and the MATRIX-PIPE
rate of the Winograd kernel (its executed FLOPs: 2 * 16 / 4 * c_in * c_out per output pixel) against the 157.3 TFLOP/s peak.
BATCH=n (default 16); WINO_ANY=1 lifts the size rule of the version (rows of shapes the rule excludes are version 1 on both sides otherwise)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import ops

SHAPES = [  # name, c_in, c_out, h, w, with gdn, with residual
    ('3x3 128->128 +gdn @272x480', 128, 128, 272, 480, True, False),
    ('3x3 128->128 +gdn @135x240', 128, 128, 135, 240, True, True),
    ('3x3 128->128 @68x120', 128, 128, 68, 120, False, True),
    ('3x3 64->64 @135x240', 64, 64, 135, 240, False, False),
    ('3x3 64->128 @135x240', 64, 128, 135, 240, False, False),
]


def main():
    dev = torch.device('cuda:0')
    nb = int(os.environ.get('BATCH', '16'))
    reps = int(os.environ.get('REPS', '10'))
    ops.WINO_ANY_SIZE = bool(os.environ.get('WINO_ANY'))  # version 2 below its size rule too (tuning aid)
    for name, ci, co, h, w, with_gdn, with_res in SHAPES:
        x = torch.randn(nb, h, w, ci, device=dev)
        wt = torch.randn(co, 3, 3, ci, device=dev) * 0.03
        b = torch.rand(co, device=dev) * 0.1
        g = (torch.rand(co, device=dev) + 0.5, torch.rand(co, co, device=dev) * 0.01, False) if with_gdn else None
        res = torch.randn(nb, h, w, co, device=dev) if with_res else None
        direct = 2.0 * 9 * ci * co * h * w * nb + (2.0 * co * co * h * w * nb if with_gdn else 0.0)
        out = []
        for mode in ('fp32', 'fp32w'):
            prev = ops.set_precision(mode)
            try:
                for _ in range(2):
                    ops.conv2d(x, wt, b, stride=1, pad=1, gdn=g, res=res)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.conv2d(x, wt, b, stride=1, pad=1, gdn=g, res=res)
                e1.record()
                torch.cuda.synchronize()
                out.append(e0.elapsed_time(e1) / reps)
                if mode == 'fp32w' and g is not None:  # the Winograd launch alone (the GDN-mode launch is the other one)
                    e0.record()
                    for _ in range(reps):
                        ops.conv2d(x, wt, b, stride=1, pad=1)
                    e1.record()
                    torch.cuda.synchronize()
                    out.append(e0.elapsed_time(e1) / reps)
            finally:
                ops.set_precision(prev)
        wino_ms = out[-1] if g is not None else out[1]
        pipe = 2.0 * 4 * ci * co * h * w * nb / (wino_ms * 1e-3) / 1e12
        print('%-30s n%-3d  v1 %7.3f ms %6.1f TF/s | v2 %7.3f ms %6.1f TF/s-equivalent  x%.2f | winograd launch %7.3f ms, matrix pipe %5.1f TF/s = %.2f of 157.3'
              % (name, nb, out[0], direct / out[0] / 1e9, out[1], direct / out[1] / 1e9, out[0] / out[1], wino_ms, pipe, pipe / 157.3))


if __name__ == '__main__':
    main()
