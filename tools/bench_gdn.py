"""Stand-alone (I)GDN: the resident-gamma kernel (csrc/gdn.hip) against the generic kernel's GDN-mode launch.
python tools/bench_gdn.py [channels] [frames] [h] [w]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aivc_amd import abi, ops  # noqa: E402


def main():
    c = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 272
    w = int(sys.argv[4]) if len(sys.argv) > 4 else 480
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((n, h, w, c), device=dev, generator=g)
    beta = torch.rand(c, device=dev, generator=g) + 0.2
    gamma = torch.rand((c, c), device=dev, generator=g) * 0.05
    out = {}
    for name, algo in (('resident', abi.ALGO_AUTO), ('generic', abi.ALGO_MFMA)):
        for inv in (False, True):
            y = ops.gdn(x, beta, gamma, inverse=inv, algo=algo)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = ops.gdn(x, beta, gamma, inverse=inv, algo=algo)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            out[(name, inv)] = (ms, y)
            gb = 2.0 * x.numel() * 4 / 1e9
            tf = 2.0 * c * c * n * h * w / 1e12
            print('%-8s %s  %7.3f ms  %6.0f GB/s  %6.1f TFLOP/s' % (name, 'igdn' if inv else 'gdn ', ms, gb / ms * 1e3, tf / ms * 1e3))
    for inv in (False, True):
        print('equal (%s): %s' % ('igdn' if inv else 'gdn', torch.equal(out[('resident', inv)][1], out[('generic', inv)][1])))


if __name__ == '__main__':
    main()
