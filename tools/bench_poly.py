"""5x5 stride-2 64 -> 128 (+ GDN) at 540x960: version 1 (tap chain, fused GDN) against the polyphase Winograd form + GDN launch.
BATCH=n (default 16)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import ops

def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def main():
    dev = torch.device('cuda:0')
    nb = int(os.environ.get('BATCH', '16'))
    for (h, w, ci, co) in [(540, 960, 64, 128), (270, 480, 64, 128), (136, 240, 64, 128)]:
        x = torch.randn(nb, h, w, ci, device=dev)
        wt = torch.randn(co, 5, 5, ci, device=dev) * 0.02
        b = torch.rand(co, device=dev) * 0.1
        g = (torch.rand(co, device=dev) + 0.5, torch.rand(co, co, device=dev) * 0.01, False)
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        direct = 2.0 * 25 * ci * co * ho * wo * nb
        out = {}
        for mode in ('fp32', 'fp32w'):
            prev = ops.set_precision(mode)
            try:
                out[mode] = (timeit(lambda: ops.conv2d(x, wt, b, stride=2, pad=2, gdn=g)), timeit(lambda: ops.conv2d(x, wt, b, stride=2, pad=2)))
            finally:
                ops.set_precision(prev)
        ex = 2.0 * 49 / 4 * ci * co * ho * wo * nb
        print('%dx%d n%d: +gdn v1 %.3f ms  v2 %.3f ms  x%.2f | conv alone v1 %.3f ms (%.1f TF/s)  polyphase %.3f ms (x%.2f; matrix pipe %.1f TF/s = %.2f)'
              % (h, w, nb, out['fp32'][0], out['fp32w'][0], out['fp32'][0] / out['fp32w'][0], out['fp32'][1], direct / out['fp32'][1] / 1e9,
                 out['fp32w'][1], out['fp32'][1] / out['fp32w'][1], ex / out['fp32w'][1] / 1e9, ex / out['fp32w'][1] / 1e9 / 157.3))

def tconv():
    from aivc_amd import abi
    dev = torch.device('cuda:0')
    nb = int(os.environ.get('BATCH', '16'))
    for (h, w, ci, co) in [(272, 480, 128, 64), (68, 120, 128, 128)]:
        x = torch.randn(nb, h, w, ci, device=dev)
        wt = torch.randn(co, 5, 5, ci, device=dev) * 0.02
        b = torch.rand(co, device=dev) * 0.1
        g = (torch.rand(co, device=dev) + 0.5, torch.rand(co, co, device=dev) * 0.01, True)
        direct = 2.0 * 6.25 * ci * co * 4 * h * w * nb
        out = {}
        for mode in ('fp32', 'fp32w'):
            prev = ops.set_precision(mode)
            try:
                out[mode] = (timeit(lambda: ops.conv2d(x, wt, b, mode=abi.MODE_TCONV, stride=2, gdn=g)), timeit(lambda: ops.conv2d(x, wt, b, mode=abi.MODE_TCONV, stride=2)))
            finally:
                ops.set_precision(prev)
        ex = 2.0 * 49 / 4 * ci * co * h * w * nb
        print('tconv5 %d->%d %dx%d n%d: +igdn v1 %.3f ms  v2 %.3f ms  x%.2f | tconv alone v1 %.3f ms (%.1f TF/s)  classes %.3f ms (x%.2f; matrix pipe %.1f TF/s = %.2f)'
              % (ci, co, h, w, nb, out['fp32'][0], out['fp32w'][0], out['fp32'][0] / out['fp32w'][0], out['fp32'][1], direct / out['fp32'][1] / 1e9,
                 out['fp32w'][1], out['fp32'][1] / out['fp32w'][1], ex / out['fp32w'][1] / 1e9, ex / out['fp32w'][1] / 1e9 / 157.3))


if __name__ == '__main__':
    main()
    tconv()
