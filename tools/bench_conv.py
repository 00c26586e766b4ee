#!/usr/bin/env python3
"""Per-layer throughput of aivc_conv2d on the shapes of the 1080p workload (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import abi, ops

SHAPES = [  # name, mode, k, s, pad, cin, cout, h_in, w_in
    ('ga0 conv5s2 12->64 @1080p', abi.MODE_CONV, 5, 2, 2, 12, 64, 1080, 1920),
    ('gdn64 @540p', abi.MODE_GDN, 1, 1, 0, 64, 64, 540, 960),
    ('ga1 conv5s2 64->128 @540p', abi.MODE_CONV, 5, 2, 2, 64, 128, 540, 960),
    ('gdn128 @270p', abi.MODE_GDN, 1, 1, 0, 128, 128, 270, 480),
    ('cheng conv3s2 128 @270p', abi.MODE_CONV, 3, 2, 1, 128, 128, 270, 480),
    ('cheng conv3 128 @135p', abi.MODE_CONV, 3, 1, 1, 128, 128, 135, 240),
    ('att 1x1 128->64 @135p', abi.MODE_CONV, 1, 1, 0, 128, 64, 135, 240),
    ('res 1x1 64->128 @135p', abi.MODE_CONV, 1, 1, 0, 64, 128, 135, 240),
    ('res 1x1 128->128 @135p', abi.MODE_CONV, 1, 1, 0, 128, 128, 135, 240),
    ('att 3x3 64->64 @135p', abi.MODE_CONV, 3, 1, 1, 64, 64, 135, 240),
    ('ga4 conv5s2 128->64 @135p', abi.MODE_CONV, 5, 2, 2, 128, 64, 135, 240),
    ('res 3x3 128 @68p', abi.MODE_CONV, 3, 1, 1, 128, 128, 68, 120),
    ('gs1 tconv5 128->128 @68p', abi.MODE_TCONV, 5, 2, 0, 128, 128, 68, 120),
    ('gs2 tconv3 128->128 @135p', abi.MODE_TCONV, 3, 2, 0, 128, 128, 135, 240),
    ('gs3 tconv5 128->64 @270p', abi.MODE_TCONV, 5, 2, 0, 128, 64, 270, 480),
    ('gs4 tconv5 64->3 @540p', abi.MODE_TCONV, 5, 2, 0, 64, 3, 540, 960),
    ('gs4 tconv5 64->6 @540p', abi.MODE_TCONV, 5, 2, 0, 64, 6, 540, 960),
]


def clock_probe():
    """Optional (CLOCK=1): side-stream spin kernel reporting the shader clock during the timed launches."""
    import ctypes, subprocess
    so = '/tmp/libclock_probe.so'
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'clock_probe.hip')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', '-w', src, '-o', so])
    lib = ctypes.CDLL(so)
    lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    side = torch.cuda.Stream()
    buf = torch.zeros(2, dtype=torch.int64, device='cuda:0')

    def start(us):
        side.wait_stream(torch.cuda.current_stream())
        lib.clock_probe_launch(buf.data_ptr(), float(us), side.cuda_stream)

    def read():
        side.synchronize()
        c, r = buf.tolist()
        return c / max(r, 1) * 0.1  # GHz (s_memrealtime = 100 MHz)
    return start, read


def main():
    dev = torch.device('cuda:0')
    clk = clock_probe() if os.environ.get('CLOCK') else None
    nb = int(os.environ.get('BATCH', '1'))
    tot_f, tot_t = 0.0, 0.0
    for name, mode, k, s, pad, ci, co, h, w in SHAPES:
        x = torch.randn(nb, h, w, ci, device=dev)
        wt = torch.randn(co, k, k, ci, device=dev) * 0.05
        b = torch.rand(co, device=dev) + 0.5
        if mode in (abi.MODE_GDN, abi.MODE_IGDN):
            wt = wt.abs()
        ho, wo = abi.conv_out_size(mode, h, w, k, s, pad)
        taps = k * k if mode != abi.MODE_TCONV else k * k / 4.0
        flops = 2.0 * taps * ci * co * ho * wo * nb
        variants = [(a, None) for a in ([abi.ALGO_AUTO] if co >= 16 else [abi.ALGO_MFMA, abi.ALGO_AUTO])]
        if co in (64, 128) and mode in (abi.MODE_CONV, abi.MODE_TCONV):
            variants.append((abi.ALGO_AUTO, (torch.rand(co, device=dev) + 0.5, torch.rand(co, co, device=dev) * 0.01, False)))
        for algo, g in variants:
            fl = flops + (2.0 * co * co * ho * wo * nb if g is not None else 0.0)
            for _ in range(2):
                ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, algo=algo, gdn=g)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            if clk:
                n = 20
                clk[0](1500.0)
            e0.record()
            for _ in range(n):
                ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, algo=algo, gdn=g)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            ops.PROFILE = []  # one more launch, to learn which kernel instantiation it was (aivc_conv2d_variant)
            ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, algo=algo, gdn=g)
            variant = ops.PROFILE[-1][0] if ops.PROFILE else -1
            ops.PROFILE = None
            print('%-32s algo=%d %s %8.3f ms  %7.1f GFLOP  %6.1f TFLOP/s  v%d' % (name, algo, '+gdn' if g is not None else '    ', ms, fl / 1e9, fl / ms / 1e9, variant)
                  + ('  clk %.3f GHz' % clk[1]() if clk else ''))
        tot_f += flops
        tot_t += ms
    print('sum: %.1f GFLOP in %.2f ms -> %.1f TFLOP/s' % (tot_f / 1e9, tot_t, tot_f / tot_t / 1e9))


if __name__ == '__main__':
    main()
