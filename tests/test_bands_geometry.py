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
    """analysis-like chain down to the y grid, a gather, a synthesis-like chain back up (tests/band_chain.py; residual /
    gate operands, odd heights, more ranks than the coarsest grid has rows): every rank's valid rows == the whole-map
    result, bit for bit"""
    import band_chain
    x, wts, H = band_chain.make_case(h_y, seed)
    t2, want = band_chain.whole(oracle, x, wts)
    assert t2.shape[1] == h_y
    res = _run_ranks(R, lambda ctx: band_chain.banded(oracle, ctx, x, wts, h_y, H) + (dict(ctx.comm.stats),))
    covered = 0
    for full, (v0, v1), rows, stats in res:
        np.testing.assert_array_equal(full, t2)
        np.testing.assert_array_equal(rows, want[:, v0:v1])
        covered += v1 - v0
    assert covered == want.shape[1] and res[0][1][0] == 0 and res[-1][1][1] == want.shape[1]
    if R > 1 and h_y >= R:
        assert max(r[3]['bytes_sent'] for r in res) > 0  # halo rows did travel
