import os, sys
sys.path.insert(0, '/root/repo')
import torch
from aivc_amd import ops
ops.WINO_ANY_SIZE = True
dev = torch.device('cuda:0')
for (h, w, nb) in [(68,120,64),(64,128,64),(64,120,64),(68,120,16),(68,120,4),(34,60,64),(34,60,16),(17,30,64),(30,52,64),(52,30,64),(135,240,64),(135,240,4),(272,480,16),(270,480,4),(100,100,16)]:
    ci = co = 128
    x = torch.randn(nb, h, w, ci, device=dev)
    wt = torch.randn(co, 3, 3, ci, device=dev) * 0.03
    b = torch.rand(co, device=dev) * 0.1
    res = torch.randn(nb, h, w, co, device=dev)
    out = []
    for mode in ('fp32', 'fp32w'):
        prev = ops.set_precision(mode)
        for _ in range(2):
            ops.conv2d(x, wt, b, stride=1, pad=1, res=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(x, wt, b, stride=1, pad=1, res=res)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10)
        ops.set_precision(prev)
    blocks = nb * ((h + 15) // 16) * ((w + 15) // 16) * 2
    pipe = 2.0 * 4 * ci * co * h * w * nb / (out[1] * 1e-3) / 1e12
    print('%3dx%3d n%-3d v1 %7.3f ms  v2 %7.3f ms  x%.2f  blocks %5d  us/block/wg %.1f  pipe %.2f' % (h, w, nb, out[0], out[1], out[0]/out[1], blocks, out[1]*1e3/(max(blocks,256)/256.0), pipe/157.3))
