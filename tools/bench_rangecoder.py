#!/usr/bin/env python3
"""Tuning aid: per-symbol cost of the range encoder and of both decoders (full CDF rows: the table-mode kernel; 64-entry
windows + scale per position: what the codec's y streams use) on Laplace-coded latents.
usage: bench_rangecoder.py   (env NSTREAMS, MAPS, SCALE)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from aivc_amd import ops


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


def under_load(fn, reps=2):
    """fn timed while the bench's dominant convolution runs back to back on another stream (LOAD=1): what the coder
    kernels get when they share the GPU with the transforms"""
    from aivc_amd import abi
    dev = torch.device('cuda:0')
    if not hasattr(under_load, 'x'):
        under_load.x = torch.randn(16, 540, 960, 64, device=dev)
        under_load.w = torch.randn(128, 5, 5, 64, device=dev) * 0.05
        under_load.b = torch.rand(128, device=dev)
        under_load.g = (torch.rand(128, device=dev) + 0.5, torch.rand(128, 128, device=dev) * 0.01, False)
        under_load.side = torch.cuda.Stream()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.stream(under_load.side):
        for _ in range(60):  # ~ 0.3 s of convolutions
            ops.conv2d(under_load.x, under_load.w, under_load.b, mode=abi.MODE_CONV, stride=2, pad=2, gdn=under_load.g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    torch.cuda.synchronize()
    return ms


def main():
    dev = torch.device('cuda:0')
    h, w, c = 68, 120, 64
    g = torch.Generator(device=dev).manual_seed(1)
    only = os.environ.get('ONLY')  # e.g. 64,64,1.5: one configuration (counter passes)
    cases = [(a, b, c_) for a in (1, 8, 64) for b in (12, 64) for c_ in (0.3, 1.5, 6.0, 30.0)]
    if only:
        a, b, c_ = only.split(',')
        cases = [(int(a), int(b), float(c_))]
    for nstreams, n_maps, sig in cases:
        if True:
            if True:
                maps = list(range(n_maps))
                sigma = (torch.rand((1, h, w, c), generator=g, device=dev) * 0.5 + 0.75) * sig
                q = (torch.randn((1, h, w, c), generator=g, device=dev) * sigma * 0.7).round().clamp(-256, 255).to(torch.int16)
                bounds = [ops.laplace_bounds(sigma, q, maps) for _ in range(nstreams)]
                nsym = n_maps * h * w
                ms_e, (out, lens, offs) = timed(lambda: ops.range_encode(bounds))
                lens_h = lens.cpu().numpy()
                out_h = out.cpu().numpy()
                payloads = [out_h[o:o + int(l)].tobytes() for (o, _), l in zip(offs, lens_h)]
                rows = torch.empty((nstreams * nsym, 520), dtype=torch.int16, device=dev)
                for i in range(nstreams):
                    ops.laplace_cdf_rows(sigma, maps, out=rows, row_off=i * nsym)
                ms_d, dec = timed(lambda: ops.range_decode(payloads, rows, [i * nsym for i in range(nstreams)], [nsym] * nstreams, [0] * nstreams))
                win, sp = ops.laplace_cdf_windows(sigma, maps)
                ms_w, dec_w = timed(lambda: ops.range_decode(payloads, win, [0] * nstreams, [nsym] * nstreams, [0] * nstreams, sigma_pos=sp))
                sym = (q[0].reshape(-1, c)[:, :n_maps].t().reshape(-1).to(torch.int32) + 256).to(torch.int16)
                if os.environ.get('LOAD') and nstreams == 64 and n_maps == 64:
                    le = under_load(lambda: ops.range_encode(bounds))
                    lw = under_load(lambda: ops.range_decode(payloads, win, [0] * nstreams, [nsym] * nstreams, [0] * nstreams, sigma_pos=sp))
                    print('   under load: encode %6.2f ms (%.3f us/sym)  decode windows %6.2f ms (%.3f us/sym)' % (le, le * 1e3 / nsym, lw, lw * 1e3 / nsym))
                ok = all(torch.equal(d, sym) for d in dec) and all(torch.equal(d, sym) for d in dec_w)
                outside = float(((q[..., :n_maps] < -32) | (q[..., :n_maps] > 30)).float().mean())
                print('streams %2d maps %2d sigma %5.1f: %6d sym/stream  %5.2f bit/sym  %4.1f %% outside the window  encode %6.2f ms (%.3f us/sym)  decode rows %6.2f ms (%.3f us/sym)  windows %6.2f ms (%.3f us/sym)  %s'
                      % (nstreams, n_maps, sig, nsym, 8.0 * lens_h.mean() / nsym, 100 * outside, ms_e, ms_e * 1e3 / nsym, ms_d, ms_d * 1e3 / nsym,
                         ms_w, ms_w * 1e3 / nsym, 'ok' if ok else 'MISMATCH'))


if __name__ == '__main__':
    main()
