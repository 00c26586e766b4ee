#!/usr/bin/env python3
"""Tuning aid: per-workgroup phase times of conv_mfma (needs the -DAIVC_TUNING build of tools/build_exp.sh).
usage: AIVC_HIP_LIB=aivc_amd/lib/exp/timing.so [BATCH=8 FUSE_GDN=1] phase_probe.py <shape-index>"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from aivc_amd import abi, ops, _lib
from conv_probe import PROBES


def main():
    idx = int(sys.argv[1])
    name, mode, k, s, pad, ci, co, h, w = PROBES[idx]
    dev = torch.device('cuda:0')
    nb = int(os.environ.get('BATCH', '1'))
    x = torch.randn(nb, h, w, ci, device=dev)
    wt = torch.randn(co, k, k, ci, device=dev) * 0.05
    b = torch.rand(co, device=dev) + 0.5
    g = None
    if os.environ.get('FUSE_GDN'):
        g = (torch.rand(co, device=dev) + 0.5, torch.rand(co, co, device=dev) * 0.01, False)
    kw = dict(gdn=g)
    if os.environ.get('TAIL'):  # fused 1x1 tail (the attention module's bottleneck block): 64 -> 128 + residual
        ho, wo = abi.conv_out_size(mode, h, w, k, s, pad)
        kw = dict(tail=(torch.randn(128, 1, 1, co, device=dev) * 0.1, torch.rand(128, device=dev)), act1=abi.ACT_LEAKY,
                  act2=abi.ACT_LEAKY, res=torch.randn(nb, ho, wo, 128, device=dev))
    if os.environ.get('RES'):  # residual + relu epilogue (the residual blocks)
        ho, wo = abi.conv_out_size(mode, h, w, k, s, pad)
        kw.update(res=torch.randn(nb, ho, wo, co, device=dev), act2=abi.ACT_NONE if g else abi.ACT_RELU)
    for _ in range(3):
        ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, **kw)
    e1.record()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    if not hasattr(lib, 'aivc_dbg_dump'):  # plain build: kernel time only (mean of 5)
        ts = []
        for _ in range(5):
            e0.record()
            ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print('shape', name, 'kernel %.1f us (min %.1f)' % (sum(ts) / len(ts), min(ts)))
        return
    n = 8 * 8192
    buf = (ctypes.c_ulonglong * n)()
    rc = lib.aivc_dbg_dump(buf, n)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    print('shape', name, 'gdn' if g else '', 'kernel %.1f us' % (e0.elapsed_time(e1) * 1e3), 'blocks', len(t), 'rc', rc)
    w0 = t[:, 0].min()
    start = (t[:, 0] - w0) / 100.0  # us
    end = (t[:, 5] - w0) / 100.0
    pro, loop, gdn, epi = t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 6] - t[:, 4]
    tot = t[:, 6] - t[:, 1]
    def st(a):
        return 'mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % (a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90))
    print('cycles  prologue+tile0 :', st(pro))
    print('cycles  main loop      :', st(loop))
    print('cycles  gdn phase      :', st(gdn))
    print('cycles  epilogue       :', st(epi))
    print('cycles  total          :', st(tot))
    print('block wall us          :', st((end - start) * 1.0), ' last end %.1f us' % end.max())
    hw = t[:, 7] & 0xffffffff
    xcc = (t[:, 7] >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    # per-CU timeline of one CU
    order = np.argsort(start)
    cu0 = cu[order[0]]
    sel = [i for i in order if cu[i] == cu0]
    print('CU %d (xcc %d) hosted %d blocks:' % (cu0 & 0xff, cu0 >> 8, len(sel)))
    for i in sel[:int(os.environ.get('ROWS', '16'))]:
        print('   blk %5d  start %8.1f us  end %8.1f us  pro %6d loop %7d gdn %6d epi %6d' % (i, start[i], end[i], pro[i], loop[i], gdn[i], epi[i]))
    ncu = len(set(cu.tolist()))
    print('distinct CUs', ncu, ' blocks/CU min %d max %d' % (np.bincount(np.unique(cu, return_inverse=True)[1]).min(), np.bincount(np.unique(cu, return_inverse=True)[1]).max()))


if __name__ == '__main__':
    main()
