import sys, time
sys.path.insert(0,'/root/repo')
import torch
from aivc_amd import synth, ops, abi
from aivc_amd.models import arch
from aivc_amd.models.conditional_net import run_nhwc
dev=torch.device('cuda:0')
model=synth.make_model(arch.DEFAULT_WIDTHS, seed=1, device=dev)
net=model.codec_net.codec_net
x=torch.randn(1,64,96,6,device=dev)   # tiny spatial size: kernels are short, CPU cost dominates
for _ in range(3): y=run_nhwc(net.g_a,x)
torch.cuda.synchronize()
n=20
t=time.time()
for _ in range(n): y=run_nhwc(net.g_a,x)
t_issue=time.time()-t
torch.cuda.synchronize()
t_all=time.time()-t
ops.PROFILE=[]
run_nhwc(net.g_a,x); 
nl=len(ops.PROFILE); ops.PROFILE=None
print('g_a: %d conv launches; CPU issue %.1f us per launch; wall %.1f us per launch'%(nl, t_issue/n/nl*1e6, t_all/n/nl*1e6))
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(10): y=run_nhwc(net.g_a,x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
