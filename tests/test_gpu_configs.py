"""BASELINE.json's configurations at their STATED workloads (frame size, frame count, coding structure), with the
synthetic default-width model (the real weights / sequences are absent):

  configs[0]  sanity_script.sh:5-13 -- 416x240, `--coding_config RA --gop_size 16 --intra_period 32`, frames 0..100
              through the aivc.py command line: 4 intra-period units of `2_GOP_16` (33 frames each, 31 of the last
              unit's are padding, src/model_mngt/model_management.py:142-153)
  configs[1]  416x240, 64 frames, all intra
  configs[2]  1280x720, 64 frames, low-delay P with intra period 8: 8 units of `LDP_8` (9 frames, 8 padded)

Checked per config: bytes of the first unit(s) == the CPU oracle's, closed loop (decoder == encoder reconstruction)
and a clean range-decoder bit count for every section of every unit, frame counts in and out.  configs[3] / [4] at
full size are in test_gpu_codec.py (one whole `1_GOP_32` unit each) and in every bench.py run."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_yuv(path, frames):
    with open(path, 'wb') as f:
        for fr in frames:
            for k in 'yuv':
                f.write(np.ascontiguousarray(fr[k]).tobytes())


def _read_yuv(path, w, h, n):
    hc, wc = (h + 1) // 2, (w + 1) // 2
    fsz = h * w + 2 * hc * wc
    raw = np.fromfile(path, np.uint8)
    assert raw.size == n * fsz, (raw.size, n * fsz)
    raw = raw.reshape(n, fsz)
    return [{'y': raw[i, :h * w].reshape(h, w), 'u': raw[i, h * w:h * w + hc * wc].reshape(hc, wc),
             'v': raw[i, h * w + hc * wc:].reshape(hc, wc)} for i in range(n)]


def _default_model(cuda):
    """what aivc_amd.cli_common.get_model builds when the assets are absent (same seeds, same calibration)"""
    from aivc_amd import synth
    model = synth.make_model(device=cuda)
    synth.calibrate_operating_point(model, cuda)
    return model


def test_config0_sanity_shape_through_the_command_line(cuda, tmp_path, capsys):
    """sanity_script.sh's invocation on a synthetic 416x240 clip (README.md:157-171 expects 26.72133 dB / 28 429 B
    from the real model + BlowingBubbles; tests/test_sanity_assets.py checks those the moment the assets exist)."""
    from aivc_amd import aivc as cli
    from aivc_amd import synth
    from aivc_amd.codec import FrameCodec
    from aivc_amd.real_life import cat_binary_files as cont
    from oracle import codec as ocodec
    from oracle import spec as ospec
    w, h, n = 416, 240, 101
    frames = synth.synthetic_video(w, h, n, seed=21)
    raw = tmp_path / ('Synthetic_%dx%d_50_420.yuv' % (w, h))
    _write_yuv(raw, frames)
    out, bits = tmp_path / 'compressed.yuv', tmp_path / 'bitstream.bin'
    status = cli.main(['-i', str(raw), '--coding_config', 'RA', '--gop_size', '16', '--intra_period', '32', '--start_frame', '0',
                       '--end_frame', '100', '--model', 'ms_ssim-2021cc-6', '--bitstream_out', str(bits), '-o', str(out)])
    printed = capsys.readouterr().out
    assert status == 0 and '[WARN]' not in printed  # every section's bit count accounts for its payload
    blob = bits.read_bytes()
    data_dim, first, last, gops = cont.unpack_video(blob)
    assert data_dim['x'] == (h, w) and (first, last) == (0, 100) and len(gops) == 4
    for g in gops:
        name, rate, fr = cont.unpack_gop(g)
        assert name == '2_GOP_16' and len(fr) == 33 and rate == 0
    for line in printed.splitlines():
        if line.startswith('Size [bytes]:'):
            assert float(line.split(':')[1]) == len(blob)
    # 101 frames out: the 31 padded frames of the last unit are dropped
    dec = _read_yuv(out, w, h, n)
    # closed loop for all 4 units: what the command line's decoder wrote == what an encoder reconstructs
    model = _default_model(cuda)
    fc = FrameCodec(model)
    with torch.no_grad():
        enc = fc.encode_video(synth.to_device_frames(frames, cuda), '2_GOP_16', idx_starting_frame=0, idx_end_frame=100)
        assert fc.assemble_video(enc) == blob  # (and the command line's bytes are the library's)
    rec = [r for g in enc['recs'] for r in g][:n]
    for i, (d, e) in enumerate(zip(dec, rec)):
        for k in 'yuv':
            np.testing.assert_array_equal(d[k], e[k][0].cpu().numpy(), err_msg='frame %d plane %s' % (i, k))
    # the first `2_GOP_16` unit against the CPU oracle: same bytes, same frames
    ref_blob, ref_rec = ocodec.encode_video(ospec.export_model(model), frames[:33], '2_GOP_16')
    _, _, _, ref_gops = cont.unpack_video(ref_blob)
    assert gops[0] == ref_gops[0]
    for i, (d, r) in enumerate(zip(dec[:33], ref_rec)):
        for k in 'yuv':
            np.testing.assert_array_equal(d[k], r[k], err_msg='frame %d plane %s' % (i, k))


def _full_config(cuda, w, h, n, gop, n_oracle_units, seed):
    from aivc_amd import synth
    from aivc_amd.codec import FrameCodec
    from aivc_amd.real_life import cat_binary_files as cont
    from oracle import codec as ocodec
    from oracle import spec as ospec
    model = _default_model(cuda)
    frames = synth.synthetic_video(w, h, n, seed=seed)
    fc = FrameCodec(model, max_batch=16)
    with torch.no_grad():
        enc = fc.encode_video(synth.to_device_frames(frames, cuda), gop)
        blob = fc.assemble_video(enc)
        dec, data_dim, first, last = fc.decode_video(blob, cuda)
    assert fc.stream_errors() == []
    assert data_dim['x'] == (h, w) and (first, last) == (0, n - 1) and len(dec) == n
    rec = [r for g in enc['recs'] for r in g][:n]
    for i, (d, e) in enumerate(zip(dec, rec)):
        for k in 'yuv':
            assert torch.equal(d[k], e[k]), (i, k)
    _, _, _, gops = cont.unpack_video(blob)
    unit = len(cont.unpack_gop(gops[0])[2])
    k = n_oracle_units * unit
    ref_blob, ref_rec = ocodec.encode_video(ospec.export_model(model), frames[:k], gop)
    _, _, _, ref_gops = cont.unpack_video(ref_blob)
    assert gops[:n_oracle_units] == ref_gops
    for i, (d, r) in enumerate(zip(dec[:k], ref_rec)):
        for p in 'yuv':
            np.testing.assert_array_equal(d[p][0].cpu().numpy(), r[p], err_msg='frame %d plane %s' % (i, p))
    return gops


def test_config1_all_intra_416x240_64_frames(cuda):
    """configs[1]: 64 I frames = 64 units of `1_GOP_0`; the first 8 against the oracle"""
    gops = _full_config(cuda, 416, 240, 64, '1_GOP_0', 8, seed=31)
    assert len(gops) == 64


def test_config2_low_delay_p_720p_64_frames(cuda):
    """configs[2]: 64 frames under `LDP_8` = 8 units of 9 (8 padded frames coded, dropped at the output); the first unit
    (I + 8 chained P frames) against the oracle"""
    gops = _full_config(cuda, 1280, 720, 64, 'LDP_8', 1, seed=32)
    assert len(gops) == 8


# ---- robustness items of the round-4 advice ---------------------------------------------------------------------------
def test_hostile_y_map_header_is_rejected_on_the_host(cuda):
    """The y section's [n_maps][map index ...] bytes come from the file and end up in a device table that indexes sigma:
    a count beyond the latent's channels, an index >= C, or a list running past the section are ContainerErrors before
    anything is uploaded (the per-frame C entry points answer AIVC_ERR_ARG to the same input)."""
    from aivc_amd import synth
    from aivc_amd.models import arch
    from aivc_amd.real_life.cat_binary_files import ContainerError
    model = synth.make_model(arch.TINY_WIDTHS, seed=11, device=cuda)
    ac = model.codec_net.codec_net.ac
    c = arch.TINY_WIDTHS['c_y']
    sigma = torch.ones((1, 4, 6, c), device=cuda)
    for bad in (b'', bytes([c + 1]) + bytes(range(c + 1)), bytes([2, 0, c]) + b'\x00' * 8, bytes([3, 0, 1]),
                bytes([255]) + bytes(255)):
        with pytest.raises(ContainerError):
            ac.decode_y([bad], sigma)
    q = ac.decode_y([b'\x00'], sigma)  # the all-zero latent is the single byte 0
    assert q.shape == (1, 4, 6, c) and not q.any()


def test_length_checks_stay_bounded_and_models_stay_picklable(cuda):
    """A caller that never asks for stream_errors() (a service, bench.py's timed loop): the queue of pending checks is
    drained as the decodes complete, the 4-byte counts use small pinned buffers that go back to the pool, and a model
    that has decoded can still be saved (the queue holds HIP events)."""
    import io
    from aivc_amd import synth
    from aivc_amd.codec import FrameCodec
    from aivc_amd.models import arch
    from aivc_amd.real_life import bitstream
    model = synth.make_model(arch.TINY_WIDTHS, seed=11, device=cuda)
    frames = synth.to_device_frames(synth.synthetic_video(64, 48, 5, noise=2.0), cuda)
    fc = FrameCodec(model)
    with torch.no_grad():
        blob = fc.assemble_video(fc.encode_video(frames, '1_GOP_4'))
        for _ in range(40):
            fc.decode_video(blob, cuda)
            torch.cuda.synchronize()
    acs = [net.ac for net in (fc.mof, fc.cod)]
    assert all(len(ac._length_checks) <= 8 for ac in acs), [len(ac._length_checks) for ac in acs]
    small = [k for k in bitstream._PIN_POOL if k[0] == torch.int32 and k[1] <= 64]
    assert small and sum(len(bitstream._PIN_POOL[k]) for k in small) <= 64
    buf = io.BytesIO()
    torch.save(model, buf)  # pending checks (events, pinned buffers) are not part of the model
    assert fc.stream_errors() == []


def test_decode_cli_status_is_returned_by_main(cuda, tmp_path, monkeypatch):
    """`main()` returns the status a process would exit with (0 clean, 3 when a section did not decode to where its
    payload ends), so console-script entry points and programmatic callers see it, not only `python -m`."""
    from aivc_amd import decode as dec_cli
    from aivc_amd import synth
    from aivc_amd.codec import FrameCodec
    from aivc_amd.real_life import cat_binary_files as cont
    from aivc_amd.real_life.bitstream import split_sections
    model = _default_model(cuda)
    monkeypatch.setattr(dec_cli, 'get_model', lambda name, dev: model)
    frames = synth.to_device_frames(synth.synthetic_video(96, 64, 3, noise=2.0), cuda)
    fc = FrameCodec(model)
    with torch.no_grad():
        blob = fc.assemble_video(fc.encode_video(frames, '1_GOP_2'))
    (tmp_path / 'ok.bin').write_bytes(blob)
    assert dec_cli.main(['-i', str(tmp_path / 'ok.bin'), '-o', str(tmp_path / 'ok.yuv')]) == 0
    f0 = cont.unpack_gop(cont.unpack_video(blob)[3][0])[2][0]
    sy = split_sections(f0)[3]
    pos = f0.index(sy) + 1 + sy[0] + (len(sy) - 1 - sy[0]) // 3
    bad_f0 = bytearray(f0)
    bad_f0[pos] ^= 0x5A
    (tmp_path / 'bad.bin').write_bytes(blob.replace(f0, bytes(bad_f0), 1))
    assert dec_cli.main(['-i', str(tmp_path / 'bad.bin'), '-o', str(tmp_path / 'bad.yuv')]) == 3
    with pytest.raises(SystemExit) as e:
        monkeypatch.setattr('sys.argv', ['decode.py', '-i', str(tmp_path / 'bad.bin'), '-o', str(tmp_path / 'bad.yuv')])
        dec_cli.cli()
    assert e.value.code == 3
    assert os.path.getsize(tmp_path / 'bad.yuv') == os.path.getsize(tmp_path / 'ok.yuv')
