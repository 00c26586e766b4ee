"""Row bands (aivc_amd/bands.py; SURVEY.md 8e / BASELINE configs[4]: one 4K unit over 8 GPUs): one frame's transforms
spread over R ranks with layer-by-layer halo exchange must reproduce the single-rank frame BIT FOR BIT -- sections
(hence bytes) and reconstructed planes -- for every frame type, odd frame sizes, more ranks than the latent has rows,
tiny and default widths (MFMA tiles, fused GDN / tail, thin output layer, aivc_conv_images) and at 3840x2160.

Here the R ranks are R threads of this process on the one GPU of the test box (bands.ThreadComm): the banded code is
the same as under torch.distributed, only the hand-over of the halo rows differs.  Real process boundaries:
tests/test_gpu_multi_process.py."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _version_1_of_the_contract(_contract_does_not_leak):
    """Row bands are bit exact because every output of version 1 of the arithmetic contract is one chain over its own
    window whatever tensor the window is cut from; a Winograd chain of version 2 (the default) depends on the tile grid and
    size of the tensor it is computed in, and FrameCodec._banded never bands there."""
    from aivc_amd import ops
    prev = ops.set_precision('fp32')
    yield
    ops.set_precision(prev)


def _run_ranks(R, fn):
    """fn(bands_ctx) in R threads -> list of results (exceptions re-raised)"""
    from aivc_amd.bands import BandCtx, ThreadComm
    shared = ThreadComm.Shared(R)
    out, err = [None] * R, []

    def work(r):
        try:
            with torch.no_grad():
                out[r] = fn(BandCtx(ThreadComm(shared, r), torch.device('cuda:0')))
        except BaseException as e:  # noqa: BLE001 -- a dead rank must not leave the others at the barrier
            err.append(e)
            shared.barrier.abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if err:
        raise [e for e in err if not isinstance(e, threading.BrokenBarrierError)][0] if any(
            not isinstance(e, threading.BrokenBarrierError) for e in err) else err[0]
    return out


def _check(model, frames, R, cuda):
    """I, P, B frames of `frames` (cur = 1, prev = 0, next = 2): banded over R ranks == one rank"""
    from aivc_amd.codec import FrameCodec
    from aivc_amd.func_util.GOP_structure import FRAME_B, FRAME_I, FRAME_P
    from aivc_amd.real_life.bitstream import finalize_frames
    fc = FrameCodec(model)
    stats = {}
    with torch.no_grad():
        ref0 = fc.encode_batch([frames[0]], [None], [None], FRAME_I)
        ref2 = fc.encode_batch([frames[2]], [ref0['rec'][0]], [None], FRAME_P)
        prev, nxt = ref0['rec'][0], ref2['rec'][0]
        for ftype, cur, p, n in ((FRAME_I, frames[0], None, None), (FRAME_P, frames[2], prev, None), (FRAME_B, frames[1], prev, nxt)):
            ref = fc.encode_batch([cur], [p], [n], ftype)
            ref_bytes = finalize_frames(ref['sections'])[0]
            outs = _run_ranks(R, lambda b: (fc.encode_banded(cur, p, n, ftype, 0., b), b))
            for r, (o, b) in enumerate(outs):
                for k in 'yuv':
                    assert torch.equal(o['rec'][0][k], ref['rec'][0][k]), (ftype, r, k)
                assert o['data_dim'] == ref['data_dim']
            assert finalize_frames(outs[0][0]['sections'])[0] == ref_bytes
            assert finalize_frames(outs[R - 1][0]['sections'])[0] == ref_bytes  # every rank holds the same latents
            # decoder side: latents from the bitstream (every rank decodes them), synthesis in bands
            yh = fc.entropy_decode([ref_bytes], ftype, ref['data_dim'], 0., cuda)
            torch.cuda.synchronize()
            dec = _run_ranks(R, lambda b: fc.synthesise_banded(yh, p, n, ftype, ref['data_dim'], b))
            for r, d in enumerate(dec):
                for k in 'yuv':
                    assert torch.equal(d[k], ref['rec'][0][k]), ('decode', ftype, r, k)
            stats[ftype] = {'launches': [b.launches for _, b in outs], 'comm': [b.comm.stats for _, b in outs]}
    return stats


@pytest.mark.parametrize('w,h,R', [(70, 50, 2), (64, 48, 3), (96, 80, 4), (33, 47, 2), (160, 112, 8)])
def test_banded_frame_equals_single_rank_tiny(w, h, R, cuda):
    from aivc_amd import synth
    from aivc_amd.models import arch
    from aivc_amd.model_mngt.model_management import attach_arithmetic_coders
    model = attach_arithmetic_coders(synth.make_model(arch.TINY_WIDTHS, seed=7, device=cuda), cuda)
    frames = synth.to_device_frames(synth.synthetic_video(w, h, 3, seed=4), cuda)
    _check(model, frames, R, cuda)


@pytest.mark.parametrize('w,h,R', [(256, 144, 2), (250, 130, 3), (416, 240, 4)])
def test_banded_frame_equals_single_rank_default_widths(w, h, R, cuda):
    """default widths: the LDS-DMA K loop, fused GDN / 1x1 tail, aivc_conv_images and the thin MFMA kernel on slabs"""
    from aivc_amd import synth
    from aivc_amd.models import arch
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=cuda)
    synth.calibrate_operating_point(model, cuda)
    frames = synth.to_device_frames(synth.synthetic_video(w, h, 3, seed=5), cuda)
    _check(model, frames, R, cuda)


def test_banded_frame_2160p_over_8_ranks(cuda):
    """BASELINE configs[4]'s frame size over the 8 ranks of one group; also the exchange volume the scheme costs"""
    from aivc_amd import synth
    from aivc_amd.func_util.GOP_structure import FRAME_B
    from aivc_amd.models import arch
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=cuda)
    synth.calibrate_operating_point(model, cuda)
    frames = synth.to_device_frames(synth.synthetic_video(3840, 2160, 3, seed=6), cuda)
    stats = _check(model, frames, 8, cuda)
    b = stats[FRAME_B]
    sent = max(c['bytes_sent'] for c in b['comm'])
    assert 0 < sent < 64 << 20  # halo rows only: tens of MB per rank and B frame, not the activations (GBs)
