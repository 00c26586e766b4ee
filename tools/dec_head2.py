"""Tuning aid: main-stream idle time at the start of a decode inside the bench's encode -> decode sequence: events on
the main stream at decode entry, before the first synthesis launch of every level, and at the end."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
from aivc_amd import synth
from aivc_amd.models import arch
from aivc_amd.codec import FrameCodec
from aivc_amd.func_util.GOP_structure import generate_gop_struct

dev = torch.device('cuda:0')
model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=dev)
synth.calibrate_operating_point(model, dev, active_y=(6, 12))
fc = FrameCodec(model, max_batch=64)
if os.environ.get('STREAMS'):
    fc.entropy_streams = int(os.environ['STREAMS'])
if os.environ.get('LOOKAHEAD'):
    fc.entropy_lookahead = int(os.environ['LOOKAHEAD'])
gop_name = '1_GOP_32'
unit = len(generate_gop_struct(gop_name))
fr = B.gpu_synthetic_unit(1920, 1080, 128, 0, dev, 666)
fr = fr + [fr[-1]] * (4 * unit - 128)
clip = [fr[u * unit:(u + 1) * unit] for u in range(4)]
orig = fc.synthesise_batch
marks = []


def hooked(*a, **k):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()  # completes when the main stream reaches this point, i.e. when everything before it is done
    r = orig(*a, **k)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    marks.append((time.perf_counter(), e0, e1))
    return r


fc.synthesise_batch = hooked
with torch.no_grad():
    for it in range(3):
        blobs, recs, dd = fc.encode_units(clip, gop_name)
        del marks[:]
        th0 = time.perf_counter()
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        fc.decode_units(blobs, dd, dev)
        th1 = time.perf_counter()
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        torch.cuda.synchronize()
        print('iter %d: decode %.1f ms (host issue %.1f ms)' % (it, start.elapsed_time(end), (th1 - th0) * 1e3))
        prev_end = start
        for li, (th, e0, e1) in enumerate(marks):
            # e0 is recorded BEFORE the wait_events of the level were issued?  no: synthesise_batch is called after
            # main.wait_event, so e0 completes when the level's latents are ready AND the previous level is done
            print('   level %d: host reached it at %6.1f ms | main stream: starts %6.1f ms, previous level ended %6.1f ms -> idle %5.1f ms, runs %6.1f ms'
                  % (li, (th - th0) * 1e3, start.elapsed_time(e0), start.elapsed_time(prev_end),
                     start.elapsed_time(e0) - start.elapsed_time(prev_end), e0.elapsed_time(e1)))
            prev_end = e1
