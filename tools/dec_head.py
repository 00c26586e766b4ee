"""Tuning aid: where does the start of a decode go?  host time of the parse, and the GPU time of the entropy stage of the
first level alone (z streams -> h_s -> windows -> y streams), against the whole decode."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
from aivc_amd import synth
from aivc_amd.models import arch
from aivc_amd.codec import FrameCodec, frame_index
from aivc_amd.real_life import cat_binary_files as container
from aivc_amd.func_util.GOP_structure import generate_gop_struct, coding_levels

dev = torch.device('cuda:0')
model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=dev)
synth.calibrate_operating_point(model, dev, active_y=(6, 12))
fc = FrameCodec(model)
gop_name = '1_GOP_32'
unit = len(generate_gop_struct(gop_name))
fr = B.gpu_synthetic_unit(1920, 1080, 128, 0, dev, 666)
fr = fr + [fr[-1]] * (4 * unit - 128)
clip = [fr[u * unit:(u + 1) * unit] for u in range(4)]
with torch.no_grad():
    for it in range(2):
        blobs, recs, dd = fc.encode_units(clip, gop_name)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        parsed = [container.unpack_gop(g) for g in blobs]
        t1 = time.perf_counter()
        gop = generate_gop_struct(gop_name)
        levels = coding_levels(gop)
        names0 = [f for f in levels[0]]
        fb = [parsed[i][2][frame_index(f)] for i in range(4) for f in names0]
        ftype = gop[names0[0]]['type']
        t2 = time.perf_counter()
        yh = fc.entropy_decode(fb, ftype, dd, 0., dev)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        dec = fc.decode_units(blobs, dd, dev)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        print('iter %d: parse %.1f ms | level-0 entropy stage: host issue %.1f ms, until done %.1f ms | whole decode %.1f ms | I-frame bytes %d'
              % (it, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3, (t5 - t4) * 1e3, len(fb[0])))
