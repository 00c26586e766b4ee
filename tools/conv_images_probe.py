#!/usr/bin/env python3
"""Run the first analysis layer (aivc_conv_images: 5x5 stride-2 conv to 64 channels + fused GDN straight from 8-bit
4:2:0 planes) a few times, for rocprofv3 --pmc / --kernel-trace passes.  usage: conv_images_probe.py <n_img> [reps]
env BATCH (default 16), H / W (default 1080 / 1920)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import ops


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    nb = int(os.environ.get('BATCH', '16'))
    h, w = int(os.environ.get('H', '1080')), int(os.environ.get('W', '1920'))
    dev = torch.device('cuda:0')
    hc, wc = (h + 1) // 2, (w + 1) // 2
    parts = [{'y': torch.randint(0, 256, (nb, h, w), dtype=torch.uint8, device=dev),
              'u': torch.randint(0, 256, (nb, hc, wc), dtype=torch.uint8, device=dev),
              'v': torch.randint(0, 256, (nb, hc, wc), dtype=torch.uint8, device=dev)} for _ in range(n_img)]
    stack = ops.ImageStack(parts, h, w, dev)
    co = 64
    wt = torch.zeros(co, 5, 5, 4 * n_img, device=dev)
    for i in range(n_img):
        wt[..., 4 * i:4 * i + 3] = torch.randn(co, 5, 5, 3, device=dev) * 0.05
    b = torch.rand(co, device=dev) + 0.5
    g = (torch.rand(co, device=dev) + 0.5, torch.rand(co, co, device=dev) * 0.01, False)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps + 1):
        if r == 1:
            ev[0].record()
        y = ops.conv2d(stack, wt, b, stride=2, pad=2, gdn=g)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    ho, wo = y.shape[1], y.shape[2]
    fl = nb * ho * wo * (2.0 * 25 * 3 * n_img * co + 2.0 * co * co)
    print('conv_images n_img=%d batch %d %dx%d: %.3f ms  %.1f TFLOP/s' % (n_img, nb, w, h, ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()
