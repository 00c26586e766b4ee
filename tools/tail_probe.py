#!/usr/bin/env python3
"""The attention modules' bottleneck block as ONE launch (3x3 64 -> 64 + leaky, fused 1x1 tail 64 -> 128 + residual +
leaky) at the 1080p workload's size.  usage: tail_probe.py [reps]; env BATCH (default 64), H / W (default 68 / 120)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import abi, ops


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    nb, h, w = int(os.environ.get('BATCH', '64')), int(os.environ.get('H', '68')), int(os.environ.get('W', '120'))
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn((nb, h, w, 64), generator=g).to(dev)
    res = torch.randn((nb, h, w, 128), generator=g).to(dev)
    wt = (torch.randn((64, 3, 3, 64), generator=g) / 24.0).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    w3 = (torch.randn((128, 1, 1, 64), generator=g) / 8.0).to(dev)
    b3 = torch.randn(128, generator=g).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps + 2):
        if r == 2:
            ev[0].record()
        y = ops.conv2d(x, wt, b, stride=1, pad=1, act1=abi.ACT_LEAKY, act2=abi.ACT_LEAKY, res=res, tail=(w3, b3))
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    fl = nb * h * w * (2.0 * 9 * 64 * 64 + 2.0 * 64 * 128)
    print('fused tail batch %d %dx%d: %.4f ms  %.1f TFLOP/s  checksum %.6e' % (nb, w, h, ms, fl / ms / 1e9, float(y.double().sum())))


if __name__ == '__main__':
    main()
