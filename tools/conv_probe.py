#!/usr/bin/env python3
"""Run ONE conv shape a few times (for rocprofv3 --pmc passes).  usage: conv_probe.py <shape-index> [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import abi, ops
from bench_conv import SHAPES

PROBES = SHAPES + [('conv3 128 @270p (38 GF)', abi.MODE_CONV, 3, 1, 1, 128, 128, 270, 480)]


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else len(PROBES) - 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    name, mode, k, s, pad, ci, co, h, w = PROBES[idx]
    dev = torch.device('cuda:0')
    nb = int(os.environ.get('BATCH', '1'))
    x = torch.randn(nb, h, w, ci, device=dev)
    wt = torch.randn(co, k, k, ci, device=dev) * 0.05
    b = torch.rand(co, device=dev) + 0.5
    if mode in (abi.MODE_GDN, abi.MODE_IGDN):
        wt = wt.abs()
    g = None
    if os.environ.get('FUSE_GDN'):
        g = (torch.rand(co, device=dev) + 0.5, torch.rand(co, co, device=dev) * 0.01, False)
    if os.environ.get('PRECISION'):  # e.g. bf16x3 (NO_PRESPLIT=1: the weights split in the K loop)
        ops.set_precision(os.environ['PRECISION'])
        ops.PRESPLIT_WEIGHTS = not os.environ.get('NO_PRESPLIT')
    for _ in range(reps):
        ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, gdn=g)
    torch.cuda.synchronize()
    print('probe', name)


if __name__ == '__main__':
    main()
