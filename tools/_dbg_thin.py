#!/usr/bin/env python3
"""debug aid: thin MFMA kernel vs the VALU thin kernel on structured inputs"""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from aivc_amd import abi, ops

dev = torch.device('cuda:0')
H, W, CI, CO, K = 6, 5, 64, 3, 5


def run(x, w, b):
    return ops.conv2d(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), mode=abi.MODE_TCONV, stride=2).cpu().numpy()


if os.environ.get('AIVC_THIN_VALU'):
    tag = 'valu'
else:
    tag = 'mfma'
res = {}
b = np.zeros(CO, np.float32)
# A: channel mapping -- x one-hot in channel c everywhere, w = (ci + 1)
w = np.zeros((CO, K, K, CI), np.float32)
w[...] = (np.arange(CI) + 1)[None, None, None, :]
for c in (0, 1, 2, 5, 17, 63):
    x = np.zeros((1, H, W, CI), np.float32); x[..., c] = 1
    y = run(x, w, b)
    res['chan%d' % c] = y[0, 4:8, 4:8, 0]
# B: spatial mapping -- one-hot pixel, all channels 1, w = 1
w1 = np.ones((CO, K, K, CI), np.float32)
for (py, px) in ((0, 0), (2, 3), (5, 4)):
    x = np.zeros((1, H, W, CI), np.float32); x[0, py, px, :] = 1
    y = run(x, w1, b)
    res['pix%d_%d' % (py, px)] = y[0, :, :, 0]
np.savez('/tmp/dbg_thin_%s.npz' % tag, **res)
if tag == 'mfma':
    subprocess.check_call([sys.executable, __file__], env=dict(os.environ, AIVC_THIN_VALU='1'))
    a, r = np.load('/tmp/dbg_thin_mfma.npz'), np.load('/tmp/dbg_thin_valu.npz')
    np.set_printoptions(linewidth=200, precision=1, suppress=True)
    for k in a.files:
        same = np.array_equal(a[k], r[k])
        print('==', k, 'same' if same else 'DIFF')
        if not same:
            print('mfma:\n', a[k]); print('valu:\n', r[k])
