"""smoke_check(): tiny encode + decode on cuda:0, compared with the CPU oracle.  Lives beside __graft_entry__.py,
OUTSIDE the product package: aivc_amd/ imports nothing from oracle/ (the oracle is test infrastructure, here only as
the checker)."""
import numpy as np
import torch


def smoke_check(width=64, height=48, gop_name='1_GOP_2', n_frames=3, verbose=True):
    from aivc_amd import synth
    from aivc_amd.models import arch
    from oracle import codec as ocodec
    from oracle import spec as ospec
    dev = torch.device('cuda:0')
    model = synth.make_model(arch.TINY_WIDTHS, seed=7, device=dev)
    frames = synth.synthetic_video(width, height, n_frames, seed=3)
    fc = model.frame_codec()
    with torch.no_grad():
        enc = fc.encode_video(synth.to_device_frames(frames, dev), gop_name)
        blob = fc.assemble_video(enc)
        dec, _, _, _ = fc.decode_video(blob, dev)
    torch.cuda.synchronize()
    ref_blob, ref_rec = ocodec.encode_video(ospec.export_model(model), frames, gop_name)
    assert blob == ref_blob, 'HIP bitstream differs from the oracle bitstream'
    for i, (d, r) in enumerate(zip(dec, ref_rec)):
        for k in 'yuv':
            assert np.array_equal(d[k][0].cpu().numpy(), r[k]), 'frame %d plane %s differs from the oracle' % (i, k)
    enc_rec = [r for g in enc['recs'] for r in g][:n_frames]
    for d, e in zip(dec, enc_rec):
        for k in 'yuv':
            assert torch.equal(d[k], e[k]), 'decoder reconstruction != encoder reconstruction'
    if verbose:
        print('smoke ok: %d frames %dx%d %s, %d bytes, bitstream == oracle, decode == oracle == encoder recon'
              % (n_frames, width, height, gop_name, len(blob)))
    return len(blob)
