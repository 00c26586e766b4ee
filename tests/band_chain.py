"""A small analysis / synthesis-like layer chain for the row-band tests (tests/test_bands_geometry.py: R threads;
tests/test_multi_process.py: R gloo processes): down two stride-2 stages with a strided 1x1 skip and a residual, an
all-gather at the coarsest grid, back up through two transposed convs with a gated 1x1 in between -- every conv mode,
stride, padding and epilogue operand the codec's transforms use, with the ORACLE's conv as the per-slab kernel."""
import numpy as np
import torch

from aivc_amd import abi
from aivc_amd.bands import Band

C, W_PIX = 4, 6


def make_case(h_y, seed):
    rng = np.random.default_rng(seed)
    H = h_y * 4 - int(rng.integers(0, 2))  # two stride-2 stages; odd heights included
    x = rng.standard_normal((1, H, W_PIX, C)).astype(np.float32)
    shapes = dict(a=(C, 5, 5, C), b=(C, 3, 3, C), c=(C, 3, 3, C), d=(C, 1, 1, C), e=(C, 5, 5, C), f=(C, 3, 3, C), g=(C, 1, 1, C), s=(C, 1, 1, C))
    wts = {n: (rng.standard_normal(s) * 0.3).astype(np.float32) for n, s in shapes.items()}
    return x, wts, H


def _conv(oracle, wts, xs, wname, mode=abi.MODE_CONV, stride=1, pad=0, res=None, mul=None, act1=0):
    return oracle.conv2d(xs, wts[wname], None, mode=mode, stride=stride, pad=pad, res=res, mul=mul, act1=act1)


def whole(oracle, x, wts):
    """-> (map at the y grid, final map)"""
    cv = lambda *a, **k: _conv(oracle, wts, *a, **k)
    t1 = cv(x, 'a', stride=2, pad=2)                              # 5x5 s2
    sk = cv(t1, 's', stride=2, pad=0)                             # 1x1 s2, unpadded (ChengResBlock skip)
    t2 = cv(cv(t1, 'b', stride=2, pad=1), 'c', pad=1, res=sk)     # 3x3 s2, 3x3 + residual -> y grid
    u1 = cv(t2, 'e', mode=abi.MODE_TCONV, stride=2)               # transposed 5x5
    g = cv(u1, 'd', act1=abi.ACT_SIGMOID, mul=u1, res=u1)         # 1x1 gate: u1 * sigmoid(.) + u1
    u2 = cv(g, 'f', mode=abi.MODE_TCONV, stride=2)                # transposed 3x3
    return t2, cv(u2, 'g')


def banded(oracle, ctx, x, wts, h_y, H):
    """this rank's part: -> (the gathered y-grid map, (v0, v1), its valid rows of the final map)"""
    ctx.set_frame(h_y, 2)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def layer(xb, wname, mode=abi.MODE_CONV, stride=1, pad=0, res=None, mul=None, act1=0):
        def launch(xs, rs, ms):
            n = lambda t: None if t is None else np.ascontiguousarray(t.numpy())
            return T(_conv(oracle, wts, n(xs), wname, mode, stride, pad, n(rs), n(ms), act1))
        return ctx.conv(launch, xb, mode, wts[wname].shape[1], stride, pad, C, res=res, mul=mul)
    o0, o1 = ctx.own(2, H)
    xb = Band(ctx, T(x[:, o0:o1]), o0, H, 2, o0, o1)  # this rank's rows of the input
    b1 = layer(xb, 'a', stride=2, pad=2)
    bs = layer(b1, 's', stride=2, pad=0)
    b2 = layer(layer(b1, 'b', stride=2, pad=1), 'c', pad=1, res=bs)
    full = ctx.gather_full(b2)
    v1 = layer(ctx.full(full, 0), 'e', mode=abi.MODE_TCONV, stride=2)
    vg = layer(v1, 'd', act1=abi.ACT_SIGMOID, mul=v1, res=v1)
    v2 = layer(vg, 'f', mode=abi.MODE_TCONV, stride=2)
    out = layer(v2, 'g')
    return full.numpy(), (out.v0, out.v1), out.rows(out.v0, out.v1).numpy().copy()
