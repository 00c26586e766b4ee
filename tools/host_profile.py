#!/usr/bin/env python3
"""Tuning aid: where the HOST spends its time in one encode + decode of the bench clip (cProfile), and how long the
main stream idles between the encoder's last kernel and the decoder's first (time.time around the two calls)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
from aivc_amd import synth  # noqa: E402
from aivc_amd.codec import FrameCodec  # noqa: E402
from aivc_amd.models import arch  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=dev)
    ay = os.environ.get('ACTIVE_Y')  # e.g. 64,64: the high-rate operating point
    synth.calibrate_operating_point(model, dev, **({'active_y': tuple(int(v) for v in ay.split(','))} if ay else {}))
    fc = FrameCodec(model, max_batch=64)
    fr = bench.gpu_synthetic_unit(1920, 1080, 128, 0, dev, 666)
    fr = fr + [fr[-1]] * 4
    units = [fr[u * 33:(u + 1) * 33] for u in range(4)]
    with torch.no_grad():
        for _ in range(2):
            blobs, recs, dd = fc.encode_units(units, '1_GOP_32')
            fc.decode_units(blobs, dd, dev)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t0 = time.time()
        pr.enable()
        blobs, recs, dd = fc.encode_units(units, '1_GOP_32')
        t1 = time.time()
        if not os.environ.get('NO_SYNC'):  # (NO_SYNC=1: as the bench runs it, the decoder issued behind the encoder)
            torch.cuda.synchronize()
        t2 = time.time()
        dec = fc.decode_units(blobs, dd, dev)
        t3 = time.time()
        torch.cuda.synchronize()
        t4 = time.time()
        pr.disable()
    print('encode: host returned after %.1f ms, GPU done after %.1f ms; decode: host %.1f ms, GPU done %.1f ms'
          % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3))
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(38)
    st.sort_stats('tottime').print_stats(22)


if __name__ == '__main__':
    main()
