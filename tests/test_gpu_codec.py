"""End-to-end parity on the GPU: HIP codec == CPU oracle codec (bitstream bytes and reconstructed
frames), decoder == encoder reconstruction, batched level-synchronous schedule == frame by frame."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cuda, w, h, n, seed=3, widths=None):
    from aivc_amd import synth
    from aivc_amd.models import arch
    model = synth.make_model(widths or arch.TINY_WIDTHS, seed=7, device=cuda)
    frames = synth.synthetic_video(w, h, n, seed=seed)
    return model, frames, synth.to_device_frames(frames, cuda)


@pytest.mark.parametrize('gop,n,w,h', [('1_GOP_0', 2, 64, 48), ('LDP_2', 4, 70, 50), ('1_GOP_8', 9, 64, 48),
                                       ('2_GOP_4', 11, 33, 47)])
def test_bitstream_and_frames_match_oracle(gop, n, w, h, cuda):
    from oracle import codec as ocodec
    from oracle import spec as ospec
    model, frames, dframes = _setup(cuda, w, h, n)
    fc = model.frame_codec()
    with torch.no_grad():
        enc = fc.encode_video(dframes, gop, idx_starting_frame=3)
        blob = fc.assemble_video(enc)
        dec, data_dim, first, last = fc.decode_video(blob, cuda)
    ref_blob, ref_rec = ocodec.encode_video(ospec.export_model(model), frames, gop, first=3)
    assert blob == ref_blob
    assert (first, last) == (3, 3 + n - 1) and len(dec) == n
    ref_dec = ocodec.decode_video(ospec.export_model(model), ref_blob)
    for d, r, r2 in zip(dec, ref_rec, ref_dec):
        for k in 'yuv':
            np.testing.assert_array_equal(d[k][0].cpu().numpy(), r[k])
            np.testing.assert_array_equal(r[k], r2[k])


@pytest.mark.parametrize('max_batch', [1, 3, 8])
def test_batched_schedule_is_byte_identical(max_batch, cuda):
    model, frames, dframes = _setup(cuda, 96, 64, 18, seed=5)
    from aivc_amd.codec import FrameCodec
    with torch.no_grad():
        ref = FrameCodec(model, max_batch=1)
        blob_ref = ref.assemble_video(ref.encode_video(dframes, '1_GOP_8'))
        fc = FrameCodec(model, max_batch=max_batch)
        enc = fc.encode_video(dframes, '1_GOP_8')
        blob = fc.assemble_video(enc)
        assert blob == blob_ref
        dec, _, _, _ = fc.decode_video(blob, cuda)
    rec = [r for g in enc['recs'] for r in g][:len(dec)]
    for d, e in zip(dec, rec):
        for k in 'yuv':
            assert torch.equal(d[k], e[k])


def test_default_width_model_closed_loop(cuda):
    """full-width synthetic model at a small frame size: decoder == encoder reconstruction and the
    frame sections decode to the encoder's symbols (the reference's in-band self-check,
    src/real_life/bitstream.py:333-350)."""
    from aivc_amd import synth
    from aivc_amd.models import arch
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=cuda)
    synth.calibrate_operating_point(model, cuda)
    frames = synth.to_device_frames(synth.synthetic_video(416, 240, 9, seed=9), cuda)
    fc = model.frame_codec()
    with torch.no_grad():
        enc = fc.encode_video(frames, '1_GOP_8')
        dec, _, _, _ = fc.decode_video(fc.assemble_video(enc), cuda)
    rec = [r for g in enc['recs'] for r in g][:len(dec)]
    for i, (d, e) in enumerate(zip(dec, rec)):
        for k in 'yuv':
            assert torch.equal(d[k], e[k]), (i, k)
