"""Row-band geometry (aivc_amd/bands.py) on the CPU: which input rows a band of outputs reads, the partition, the
exchange plan, slab alignment and the valid-row bookkeeping -- checked by pushing random layer chains through
BandCtx.conv with the ORACLE's conv as the per-slab kernel (R virtual ranks = R threads, host tensors) and comparing
every rank's band with the same chain on the whole map, bit for bit.  The GPU twins (HIP kernels, whole codec, real
processes) are tests/test_gpu_bands.py and tests/test_gpu_multi_process.py."""
import threading

import numpy as np
import pytest
import torch

from aivc_amd import abi
from aivc_amd.bands import Band, BandCtx, ThreadComm, need_rows


def test_need_rows():
    # replicate-padded conv, src/layers/misc/custom_conv_layers.py:145-153
    assert need_rows(abi.MODE_CONV, 5, 2, 2, 3, 7, 100) == (4, 15)
    assert need_rows(abi.MODE_CONV, 5, 2, 2, 0, 2, 100) == (0, 5)      # top edge: the clamp is the padding
    assert need_rows(abi.MODE_CONV, 3, 1, 1, 98, 100, 100) == (97, 100)  # bottom edge
    assert need_rows(abi.MODE_CONV, 1, 2, 0, 4, 6, 100) == (8, 11)     # unpadded 1x1 stride 2 (ChengResBlock skip, :75)
    # transposed conv k, stride 2, padding (k + 1) / 2 - 1, output_padding 1 (:206-223): output o reads i = (o + tp - ky) / 2
    assert need_rows(abi.MODE_TCONV, 5, 2, 0, 6, 14, 100) == (2, 8)
    assert need_rows(abi.MODE_TCONV, 3, 2, 0, 6, 14, 100) == (3, 8)
    assert need_rows(abi.MODE_TCONV, 5, 2, 0, 0, 4, 10) == (0, 3)
    assert need_rows(abi.MODE_TCONV, 5, 2, 0, 16, 20, 10) == (7, 10)
    assert need_rows(abi.MODE_CONV, 3, 1, 1, 5, 5, 100) == (0, 0)      # an empty band reads nothing
    # brute force: every (mode, k, stride, pad) of the codec against the defining sums
    for mode, k, s, p in ((abi.MODE_CONV, 5, 2, 2), (abi.MODE_CONV, 3, 2, 1), (abi.MODE_CONV, 3, 1, 1), (abi.MODE_CONV, 5, 1, 2),
                          (abi.MODE_CONV, 1, 1, 0), (abi.MODE_CONV, 1, 2, 0), (abi.MODE_TCONV, 5, 2, 0), (abi.MODE_TCONV, 3, 2, 0)):
        for h in (1, 2, 7, 12):
            h_out = (h + 2 * p - k) // s + 1 if mode == abi.MODE_CONV else 2 * h
            for o0 in range(h_out):
                for o1 in range(o0 + 1, h_out + 1):
                    rows = set()
                    for o in range(o0, o1):
                        for ky in range(k):
                            if mode == abi.MODE_CONV:
                                rows.add(min(max(s * o - p + ky, 0), h - 1))
                            else:
                                num = o + (k + 1) // 2 - 1 - ky
                                if num % 2 == 0 and 0 <= num // 2 < h:
                                    rows.add(num // 2)
                    lo, hi = need_rows(mode, k, s, p, o0, o1, h)
                    assert rows <= set(range(lo, hi)), (mode, k, s, p, h, o0, o1)
                    assert not rows or (min(rows) == lo and max(rows) == hi - 1), (mode, k, s, p, h, o0, o1, lo, hi, rows)


def _run_ranks(R, fn):
    shared = ThreadComm.Shared(R)
    out, err = [None] * R, []

    def work(r):
        try:
            out[r] = fn(BandCtx(ThreadComm(shared, r), torch.device('cpu')))
        except BaseException as e:  # noqa: BLE001
            err.append(e)
            shared.barrier.abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if err:
        raise next((e for e in err if not isinstance(e, threading.BrokenBarrierError)), err[0])
    return out


@pytest.mark.parametrize('h_y,R,seed', [(3, 2, 0), (5, 3, 1), (2, 4, 2), (9, 4, 3), (7, 8, 4), (17, 8, 5)])
def test_banded_chain_equals_whole_map(h_y, R, seed, oracle):
    """analysis-like chain down to the y grid, a gather, a synthesis-like chain back up (with residual / gate operands
    and an odd top height): every rank's valid rows == the whole-map result"""
    rng = np.random.default_rng(seed)
    c, w = 4, 6
    H = h_y * 4 - int(rng.integers(0, 2))  # two stride-2 stages; odd sizes included
    x = rng.standard_normal((1, H, w, c)).astype(np.float32)
    W = {n: (rng.standard_normal(s) * 0.3).astype(np.float32) for n, s in
         dict(a=(c, 5, 5, c), b=(c, 3, 3, c), c=(c, 3, 3, c), d=(c, 1, 1, c), e=(c, 5, 5, c), f=(c, 3, 3, c), g=(c, 1, 1, c), s=(c, 1, 1, c)).items()}

    def conv(xs, wname, mode=abi.MODE_CONV, stride=1, pad=0, res=None, mul=None, act1=0):
        return oracle.conv2d(xs, W[wname], None, mode=mode, stride=stride, pad=pad, res=res, mul=mul, act1=act1)

    # whole map
    t1 = conv(x, 'a', stride=2, pad=2)                       # 5x5 s2
    sk = conv(t1, 's', stride=2, pad=0)                      # 1x1 s2 unpadded skip
    t2 = conv(conv(t1, 'b', stride=2, pad=1), 'c', pad=1, res=sk)  # 3x3 s2, 3x3 + residual -> y grid
    assert t2.shape[1] == h_y
    u1 = conv(t2, 'e', mode=abi.MODE_TCONV, stride=2)        # tconv 5
    g = conv(u1, 'd', act1=abi.ACT_SIGMOID, mul=u1, res=u1)  # 1x1 gate: u1 * sigmoid(.) + u1
    u2 = conv(g, 'f', mode=abi.MODE_TCONV, stride=2)         # tconv 3
    want = conv(u2, 'g')

    def rank(ctx):
        ctx.set_frame(h_y, 2)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))

        def layer(xb, wname, mode=abi.MODE_CONV, stride=1, pad=0, res=None, mul=None, act1=0):
            def launch(xs, rs, ms):
                n = lambda t: None if t is None else np.ascontiguousarray(t.numpy())
                return T(conv(n(xs), wname, mode, stride, pad, n(rs), n(ms), act1))
            return ctx.conv(launch, xb, mode, W[wname].shape[1], stride, pad, c, res=res, mul=mul)
        o0, o1 = ctx.own(2, H)
        xb = Band(ctx, T(x[:, o0:o1]), o0, H, 2, o0, o1)  # this rank's rows of the input
        b1 = layer(xb, 'a', stride=2, pad=2)
        bs = layer(b1, 's', stride=2, pad=0)
        b2 = layer(layer(b1, 'b', stride=2, pad=1), 'c', pad=1, res=bs)
        full = ctx.gather_full(b2)
        np.testing.assert_array_equal(full.numpy(), t2)
        v1 = layer(ctx.full(full, 0), 'e', mode=abi.MODE_TCONV, stride=2)
        vg = layer(v1, 'd', act1=abi.ACT_SIGMOID, mul=v1, res=v1)
        v2 = layer(vg, 'f', mode=abi.MODE_TCONV, stride=2)
        out = layer(v2, 'g')
        return out.v0, out.v1, out.rows(out.v0, out.v1).numpy().copy(), dict(ctx.comm.stats)
    res = _run_ranks(R, rank)
    covered = 0
    for v0, v1, rows, stats in res:
        np.testing.assert_array_equal(rows, want[:, v0:v1])
        covered += v1 - v0
    assert covered == want.shape[1] and res[0][0] == 0 and res[-1][1] == want.shape[1]
    if R > 1 and h_y >= R:
        assert max(s['bytes_sent'] for _, _, _, s in res) > 0  # halo rows did travel
