#!/usr/bin/env python3
"""Tuning aid: the thin output layer (transposed 5x5, 64 -> 3 / 6) over batch sizes (GPU box)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import abi, ops

dev = torch.device('cuda:0')
h, w = 544, 960
for co in (3, 6):
    wt = torch.randn(co, 5, 5, 64, device=dev) * 0.05
    b = torch.rand(co, device=dev)
    for n in [int(v) for v in os.environ.get('BATCHES', '4,8,16,32,48,64').split(',')]:
        x = torch.randn(n, h, w, 64, device=dev)
        for _ in range(2):
            y = ops.conv2d(x, wt, b, mode=abi.MODE_TCONV, stride=2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = ops.conv2d(x, wt, b, mode=abi.MODE_TCONV, stride=2)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.0 * 25 / 4 * 64 * co * 4 * h * w * n
        print('64->%d n=%2d %7.3f ms  %6.1f TFLOP/s  %6.1f us/frame' % (co, n, ms, fl / ms / 1e9, ms / n * 1e3))
        del x, y
