"""Tuning aid: main-stream time of the encoder's transforms per dependency level, pipelined (range coding of the
previous levels on the side streams) and with every coder launch held back to the end (DEFER=1)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
from aivc_amd import synth, codec
from aivc_amd.models import arch
from aivc_amd.codec import FrameCodec
from aivc_amd.func_util.GOP_structure import generate_gop_struct

dev = torch.device('cuda:0')
model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=dev)
synth.calibrate_operating_point(model, dev, active_y=(6, 12))
fc = FrameCodec(model, max_batch=64)
gop_name = '1_GOP_32'
unit = len(generate_gop_struct(gop_name))
fr = B.gpu_synthetic_unit(1920, 1080, 128, 0, dev, 666)
fr = fr + [fr[-1]] * (4 * unit - 128)
clip = [fr[u * unit:(u + 1) * unit] for u in range(4)]
orig = fc.encode_batch
marks = []


def hooked(*a, **k):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(*a, **k)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    marks.append((e0, e1))
    return r


fc.encode_batch = hooked
if os.environ.get('DEFER'):
    real = codec.launch_finalize
    held = []

    class Late:
        def __init__(self, args, kw):
            self.args, self.kw, self.job = args, kw, None

        def collect(self):
            for h in held:  # first collect: launch everything that was held back
                if h.job is None:
                    h.job = real(*h.args, **h.kw)
            return self.job.collect()

    def lazy(*a, **k):
        h = Late(a, k)
        held.append(h)
        return h
    codec.launch_finalize = lazy
with torch.no_grad():
    for it in range(3):
        del marks[:]
        if os.environ.get('DEFER'):
            del held[:]
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        t0 = time.perf_counter()
        blobs, recs, dd = fc.encode_units(clip, gop_name)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print('iter %d: encode %.1f ms wall | per level on the main stream (ms): %s  sum %.1f | last level ends at %.1f ms'
              % (it, (t1 - t0) * 1e3, ' '.join('%.1f' % a.elapsed_time(b) for a, b in marks),
                 sum(a.elapsed_time(b) for a, b in marks), start.elapsed_time(marks[-1][1])))
