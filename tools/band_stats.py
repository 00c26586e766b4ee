#!/usr/bin/env python3
"""Row bands (aivc_amd/bands.py) in numbers, on ONE GPU: R virtual ranks (threads) code an I, a P and a B frame in
bands; per frame type and rank: kernel launches, halo exchanges, bytes sent point-to-point, bytes all-gathered; and the
GPU time of the whole banded frame (all R bands back to back on the one GPU = the scheme's total work) next to the
single-rank frame, i.e. what the slabs, partial tiles and small launches cost.

    python tools/band_stats.py [--width 3840 --height 2160 --ranks 8]
"""
import argparse
import json
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def run_ranks(R, fn):
    from aivc_amd.bands import BandCtx, ThreadComm
    shared = ThreadComm.Shared(R)
    out = [None] * R

    def work(r):
        with torch.no_grad():
            out[r] = fn(BandCtx(ThreadComm(shared, r), torch.device('cuda:0')))
    ts = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return out


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, r


def kernel_ms(fn):
    """sum of the GPU kernel durations of fn() (the thread simulation is host bound -- 8 x ~100 launches with barriers
    under one GIL -- so wall time says nothing about the bands' GPU work)"""
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    tot = 0.0
    for e in prof.events():
        if e.device_type is not None and 'cuda' in str(e.device_type).lower():
            tot += e.device_time if hasattr(e, 'device_time') else e.cuda_time
    return tot / 1e3


def kernel_table(fn, top=14):
    """{kernel name (short): (calls, ms)} of fn()'s GPU kernels, largest first"""
    import collections
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    per = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type is not None and 'cuda' in str(e.device_type).lower():
            k = e.name.split('(')[0].replace('void ', '').replace('aivc::', '')[:70]
            per[k][0] += 1
            per[k][1] += (e.device_time if hasattr(e, 'device_time') else e.cuda_time) / 1e3
    return sorted(per.items(), key=lambda kv: -kv[1][1])[:top]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--width', type=int, default=3840)
    ap.add_argument('--height', type=int, default=2160)
    ap.add_argument('--ranks', type=int, default=8)
    ap.add_argument('--detail', action='store_true', help='per-kernel tables of the B frame, single rank vs all bands (stderr)')
    a = ap.parse_args()
    from aivc_amd import synth
    from aivc_amd.codec import FrameCodec
    from aivc_amd.func_util.GOP_structure import FRAME_B, FRAME_I, FRAME_P
    from aivc_amd.models import arch
    dev = torch.device('cuda:0')
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=dev)
    synth.calibrate_operating_point(model, dev)
    frames = synth.to_device_frames(synth.synthetic_video(a.width, a.height, 3, seed=6), dev)
    fc = FrameCodec(model)
    out = {'frame': '%dx%d' % (a.width, a.height), 'ranks': a.ranks, 'per_frame_type': {}}
    with torch.no_grad():
        r0 = fc.encode_batch([frames[0]], [None], [None], FRAME_I)
        r2 = fc.encode_batch([frames[2]], [r0['rec'][0]], [None], FRAME_P)
        prev, nxt = r0['rec'][0], r2['rec'][0]
        for name, ftype, cur, p, n in (('I', FRAME_I, frames[0], None, None), ('P', FRAME_P, frames[2], prev, None), ('B', FRAME_B, frames[1], prev, nxt)):
            t_one, _ = timed(lambda: fc.encode_batch([cur], [p], [n], ftype))
            t_all, res = timed(lambda: run_ranks(a.ranks, lambda b: (fc.encode_banded(cur, p, n, ftype, 0., b), b)))
            st = [b.comm.stats for _, b in res]
            k_one = kernel_ms(lambda: fc.encode_batch([cur], [p], [n], ftype))
            k_all = kernel_ms(lambda: run_ranks(a.ranks, lambda b: fc.encode_banded(cur, p, n, ftype, 0., b)))
            out['per_frame_type'][name] = {
                'encode_ms_single_rank': round(t_one, 2), 'encode_ms_all_bands_on_one_gpu': round(t_all, 2),
                'note': 'the two wall times above are HOST bound in the simulation; kernel_ms_* are sums of GPU kernel durations',
                'kernel_ms_single_rank': round(k_one, 2), 'kernel_ms_all_bands': round(k_all, 2),
                'gpu_work_ratio': round(k_all / k_one, 3), 'kernel_ms_per_rank': round(k_all / a.ranks, 2),
                'launches_per_rank': [b.launches for _, b in res],
                'exchanges_per_rank': max(s['exchanges'] for s in st),
                'p2p_bytes_sent_per_rank_max': max(s['bytes_sent'] for s in st),
                'all_gathers_per_rank': max(s['gathers'] for s in st),
                'all_gather_bytes_received_per_rank': max(s['bytes_gathered'] for s in st)}
    if a.detail:
        with torch.no_grad():
            cur, p, n = frames[1], prev, nxt
            for title, fn in (('single rank', lambda: fc.encode_batch([cur], [p], [n], FRAME_B)),
                              ('all %d bands' % a.ranks, lambda: run_ranks(a.ranks, lambda b: fc.encode_banded(cur, p, n, FRAME_B, 0., b)))):
                sys.stderr.write('--- B frame, %s\n' % title)
                for k, (c, ms) in kernel_table(fn):
                    sys.stderr.write('%6d x %8.3f ms  %s\n' % (c, ms, k))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
