#!/usr/bin/env python3
"""Tuning aid: the precision mode's kernels on the bench's big layers (env AIVC_BF16X3_TILE selects the tile variant,
ops.PRESPLIT_WEIGHTS the split).  Prints ms per launch and TFLOP/s fp32-equivalent."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import abi, ops

SHAPES = [('ga1 conv5s2 64->128+gdn @540p', abi.MODE_CONV, 5, 2, 2, 64, 128, 540, 960, 16, 1),
          ('cheng conv3 128 @135p', abi.MODE_CONV, 3, 1, 1, 128, 128, 135, 240, 64, 0),
          ('res 3x3 128 @68p', abi.MODE_CONV, 3, 1, 1, 128, 128, 68, 120, 64, 0),
          ('ga4 conv5s2 128->64 @135p', abi.MODE_CONV, 5, 2, 2, 128, 64, 135, 240, 64, 0),
          ('gs1 tconv5 128->128+igdn @68p', abi.MODE_TCONV, 5, 2, 0, 128, 128, 68, 120, 64, 2),
          ('gs3 tconv5 128->64+igdn @270p', abi.MODE_TCONV, 5, 2, 0, 128, 64, 270, 480, 16, 2)]


def main():
    dev = torch.device('cuda:0')
    ops.set_precision('bf16x3')
    ops.PRESPLIT_WEIGHTS = not os.environ.get('NO_PRESPLIT')
    for name, mode, k, s, pad, ci, co, h, w, nb, gdn in SHAPES:
        x = torch.randn(nb, h, w, ci, device=dev)
        wt = torch.randn(co, k, k, ci, device=dev) * 0.05
        b = torch.rand(co, device=dev) + 0.5
        g = (torch.rand(co, device=dev) + 0.5, torch.rand(co, co, device=dev) * 0.01, gdn == 2) if gdn else None
        for _ in range(2):
            y = ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, gdn=g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 5
        for _ in range(reps):
            y = ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, gdn=g)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        pix = nb * h * w if mode == abi.MODE_TCONV else y.shape[0] * y.shape[1] * y.shape[2]
        fl = 2.0 * k * k * ci * co * pix + (2.0 * co * co * y.shape[0] * y.shape[1] * y.shape[2] if gdn else 0)
        print('%-34s %8.3f ms %7.1f TFLOP/s' % (name, ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()
