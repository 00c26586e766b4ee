"""Tuning aid: main-stream time of the decoder's synthesis per dependency level with the entropy stage of EVERY level
finished beforehand (nothing on the side streams), against the pipelined decode (tools/dec_head2.py)."""
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
fc.entropy_lookahead = 99  # every level's entropy stage is issued before the first synthesis
gop_name = '1_GOP_32'
unit = len(generate_gop_struct(gop_name))
fr = B.gpu_synthetic_unit(1920, 1080, 128, 0, dev, 666)
fr = fr + [fr[-1]] * (4 * unit - 128)
clip = [fr[u * unit:(u + 1) * unit] for u in range(4)]
orig = fc.synthesise_batch
marks = []
first = [True]


def hooked(*a, **k):
    if first[0]:
        torch.cuda.synchronize()  # all entropy stages done: the side streams are empty from here on
        first[0] = False
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(*a, **k)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    marks.append((e0, e1))
    return r


fc.synthesise_batch = hooked
with torch.no_grad():
    for it in range(3):
        blobs, recs, dd = fc.encode_units(clip, gop_name)
        del marks[:]
        first[0] = True
        fc.decode_units(blobs, dd, dev)
        torch.cuda.synchronize()
        print('iter %d: synthesis alone per level (ms): %s  sum %.1f' % (it, ' '.join('%.1f' % a.elapsed_time(b) for a, b in marks),
                                                                        sum(a.elapsed_time(b) for a, b in marks)))
