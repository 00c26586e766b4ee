import sys, time, os
sys.path.insert(0,'/root/repo')
import torch
from aivc_amd import synth
from aivc_amd.codec import FrameCodec
from aivc_amd.models import arch
import aivc_amd.codec as codec_mod
import aivc_amd.real_life.bitstream as bs
from aivc_amd import ops
dev=torch.device('cuda:0')
model=synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=dev)
synth.calibrate_operating_point(model, dev)
sys.path.insert(0,'/root/repo')
from bench import gpu_synthetic_unit
clip=gpu_synthetic_unit(1920,1080,132,0,dev,666)
fc=FrameCodec(model)
with torch.no_grad():
    enc=fc.encode_video(clip,'1_GOP_32'); blob=fc.assemble_video(enc)
    fc.decode_video(blob,dev); torch.cuda.synchronize()
    # time host-side pieces
    T={}
    def wrap(mod,name):
        f=getattr(mod,name)
        def g(*a,**k):
            t=time.perf_counter(); r=f(*a,**k); T[name]=T.get(name,0)+time.perf_counter()-t; return r
        setattr(mod,name,g)
    for n in ('range_decode','laplace_cdf_rows','scatter_symbols','conv2d','dequantize','hyper_params','warp_blend','frame_to_yuv420','yuv420_to_444'):
        wrap(ops,n)
    orig_ed=FrameCodec.entropy_decode; orig_sb=FrameCodec.synthesise_batch
    def ed(self,*a,**k):
        t=time.perf_counter(); r=orig_ed(self,*a,**k); T['entropy_decode']=T.get('entropy_decode',0)+time.perf_counter()-t; return r
    def sb(self,*a,**k):
        t=time.perf_counter(); r=orig_sb(self,*a,**k); T['synthesise_batch']=T.get('synthesise_batch',0)+time.perf_counter()-t; return r
    FrameCodec.entropy_decode=ed; FrameCodec.synthesise_batch=sb
    t=time.perf_counter()
    fc.decode_video(blob,dev)
    t_issue=time.perf_counter()-t
    torch.cuda.synchronize()
    t_all=time.perf_counter()-t
print('decode: host issue %.3f s, total %.3f s'%(t_issue,t_all))
for k,v in sorted(T.items(), key=lambda kv:-kv[1]): print('  %-20s %.3f s'%(k,v))
