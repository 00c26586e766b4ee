"""CPU tier: version 2 of the fp32 arithmetic contract in the ORACLE (oracle/aivc_oracle.c) -- the Winograd chains of the three layer
classes it covers (stride-1 3x3, 5x5 stride 2 in polyphase form, transposed 5x5 stride 2 class by class; include/aivc_hip.h,
aivc_winograd_covers) against an fp64 evaluation of the same layers with torch, next to version 1's error; the weight transforms
against G g G^T in fp64; the structural zeros of the 5x5 forms; the default contract and its environment switch.  (HIP == oracle bit for
bit is the GPU tier's tests/test_gpu_winograd.py.)"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from aivc_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def both(oracle):
    prev = oracle.set_precision('fp32')
    oracle.WINO_ANY_SIZE = True
    yield oracle
    oracle.WINO_ANY_SIZE = False
    oracle.set_precision(prev)


def _ref(kind, x, w, b):
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    if kind == 'conv3':
        y = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, (1, 1, 1, 1), mode='replicate'), torch.from_numpy(w).double().permute(0, 3, 1, 2), torch.from_numpy(b).double())
    elif kind == 'conv5s2':
        y = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, (2, 2, 2, 2), mode='replicate'), torch.from_numpy(w).double().permute(0, 3, 1, 2), torch.from_numpy(b).double(), stride=2)
    else:  # transposed 5x5 stride 2: w[co][ky][kx][ci] = torch weight[ci][co][ky][kx]; padding int((1 + k) / 2 - 1) = 2, output_padding 1
        y = torch.nn.functional.conv_transpose2d(xt, torch.from_numpy(w).double().permute(3, 0, 1, 2), torch.from_numpy(b).double(), stride=2, padding=2, output_padding=1)
    return y.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize('kind,shape', [('conv3', (1, 13, 17, 32, 128)), ('conv5s2', (2, 15, 18, 32, 128)), ('tconv5', (1, 9, 11, 32, 64))])
def test_oracle_version_2_against_fp64_next_to_version_1(kind, shape, both):
    orc = both
    n, h, w, ci, co = shape
    rng = np.random.default_rng(hash(kind) % 1000)
    k = 3 if kind == 'conv3' else 5
    x = (rng.standard_normal((n, h, w, ci)) * 2).astype(np.float32)
    wt = (rng.standard_normal((co, k, k, ci)) / np.sqrt(k * k * ci)).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    ref = _ref(kind, x, wt, b)
    kw = dict(stride=1, pad=1) if kind == 'conv3' else (dict(stride=2, pad=2) if kind == 'conv5s2' else dict(mode=abi.MODE_TCONV, stride=2))
    errs = {}
    for mode in ('fp32', 'fp32w'):
        orc.set_precision(mode)
        y = orc.conv2d(x, wt, b, **kw)
        assert y.shape == ref.shape
        e = np.abs(y - ref) / np.maximum(1.0, np.abs(ref))
        errs[mode] = (float(e.max()), float(np.sqrt((e ** 2).mean())), y)
    assert not np.array_equal(errs['fp32'][2], errs['fp32w'][2])  # (the layer IS covered: another chain, other bits)
    assert errs['fp32w'][0] <= max(2e-5, 4 * errs['fp32'][0]) and errs['fp32w'][1] <= 3 * errs['fp32'][1], (kind, errs['fp32'][:2], errs['fp32w'][:2])


def test_oracle_weight_transforms(oracle):
    rng = np.random.default_rng(3)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    w5 = (rng.standard_normal((64, 5, 5, 32)) * 2).astype(np.float32)
    # polyphase form: phase (py, px) kernel g[r][l] = w[2 r + py][2 l + px] (0 beyond the 5x5), 4 * 32 virtual input channels
    u = oracle.winograd_weights(w5).reshape(1, 16, 16, 2, 64, 4)  # [co / 64][cv / 8][p][(cv % 8) / 4][co % 64][cv % 4]
    back = np.transpose(u, (0, 4, 2, 1, 3, 5)).reshape(64, 16, 128)
    for phase in range(4):
        py, px = phase >> 1, phase & 1
        g = np.zeros((64, 3, 3, 32))
        for r in range(3):
            for l in range(3):
                if 2 * r + py < 5 and 2 * l + px < 5:
                    g[:, r, l] = w5[:, 2 * r + py, 2 * l + px]
        ref = np.einsum('ik,oklc,jl->oijc', G, g, G).reshape(64, 16, 32)
        got = back[:, :, 32 * phase:32 * phase + 32]
        assert np.abs(got - ref).max() <= np.abs(ref).max() * 2.0 ** -23
        for pos in range(16):
            assert (np.abs(got[:, pos]).max() == 0.0) == ((py == 1 and pos >> 2 == 3) or (px == 1 and pos & 3 == 3))
    # transposed form: class (pyc, pxc) kernel g[r][l] = w[pyc + 4 - 2 r][pxc + 4 - 2 l], 4 * 64 virtual output channels
    ut = oracle.winograd_weights(w5, transposed=True).reshape(4, 4, 16, 2, 64, 4)  # [class][ci / 8][p][(ci % 8) / 4][co][ci % 4]
    for cls in range(4):
        pyc, pxc = cls >> 1, cls & 1
        g = np.zeros((64, 3, 3, 32))
        for r in range(3):
            for l in range(3):
                if pyc + 4 - 2 * r < 5 and pxc + 4 - 2 * l < 5:
                    g[:, r, l] = w5[:, pyc + 4 - 2 * r, pxc + 4 - 2 * l]
        ref = np.einsum('ik,oklc,jl->oijc', G, g, G).reshape(64, 16, 32)
        got = np.transpose(ut[cls], (3, 1, 0, 2, 4)).reshape(64, 16, 32)
        assert np.abs(got - ref).max() <= np.abs(ref).max() * 2.0 ** -23
        for pos in range(16):
            assert (np.abs(got[:, pos]).max() == 0.0) == ((pyc == 1 and pos >> 2 == 0) or (pxc == 1 and pos & 3 == 0))


def test_default_contract_and_its_environment_switch():
    code = "from aivc_amd import ops, abi; from oracle import oracle as o; print(ops.DEFAULT_CONTRACT, ops.PRECISION == abi.PREC_FP32_WINO, o.PRECISION == ops.PRECISION)"
    for env_val, want in ((None, 'fp32w True True'), ('fp32', 'fp32 False True'), ('fp32w', 'fp32w True True')):
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        env.pop('AIVC_CONTRACT', None)
        if env_val:
            env['AIVC_CONTRACT'] = env_val
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.strip().splitlines()[-1] == want, (env_val, r.stdout)
