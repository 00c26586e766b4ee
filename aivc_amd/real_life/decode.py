"""Frame / GOP / video decoder with the reference's class and function names
(src/real_life/decode.py).  Decoder.decode keeps the reference's dictionary contract (float YUV
dicts, one bitstream file per frame); decode_one_video reads the container and writes planar YUV
directly (the reference's PNG triplets + per-frame `dd` forks are out of scope)."""
import os
import time

import numpy as np
import torch
from torch.nn import Module

from ..codec import FrameCodec
from ..func_util.console_display import print_log_msg
from ..func_util.nn_util import get_value
from ..models.full_net import _to_float_dic, _to_u8_planes
from .utils import BITSTREAM_SUFFIX


class ConditionalDecoder(Module):
    """Decoder half of a ConditionalNet (src/real_life/decode.py:752-898): shares the sub-modules."""

    def __init__(self, param):
        super().__init__()
        net = get_value('conditional_net', param, {'conditional_net': None})
        self.net = net
        self.g_s, self.h_s = net.g_s, net.h_s
        self.g_a_ref = getattr(net, 'g_a_ref', None)
        self.pdf_y, self.pdf_z, self.pdf_parameterizer = net.pdf_y, net.pdf_z, net.pdf_parameterizer
        self.nb_ft_shortcut_out, self.nb_ft_y, self.nb_ft_z = net.out_c_shortcut_y, net.nb_ft_y, net.nb_ft_z
        self.gain_I, self.flag_gain_p_b = net.gain_I, net.flag_gain_p_b
        if self.flag_gain_p_b:
            self.gain_P, self.gain_B = net.gain_P, net.gain_B
        self.ac = net.ac


class CodecNetDecoder(Module):
    def __init__(self, param):
        super().__init__()
        self.codec_dec = ConditionalDecoder({'conditional_net': get_value('codec_net', param, {'codec_net': None}).codec_net})


class MOFNetDecoder(Module):
    def __init__(self, param):
        super().__init__()
        self.mofnet_dec = ConditionalDecoder({'conditional_net': get_value('mofnet', param, {'mofnet': None}).mode_net})


class Decoder(Module):
    """Entire decoder, built from a complete FullNet (src/real_life/decode.py:429-580)."""

    def __init__(self, param):
        super().__init__()
        full_net = get_value('full_net', param, {'full_net': None})
        self.full_net = full_net
        self.codec_net_dec = CodecNetDecoder({'codec_net': full_net.codec_net})
        self.mofnet_dec = MOFNetDecoder({'mofnet': full_net.mode_net})
        self.motion_compensation = full_net.motion_compensation
        self.in_layer, self.out_layer = full_net.in_layer, full_net.out_layer

    def decode(self, param):
        """{'prev_dic','next_dic','frame_type','bitstream_path','data_dim','idx_rate','device'} ->
        decoded frame as a float YUV dict holding 8-bit levels."""
        default = {'prev_dic': None, 'next_dic': None, 'frame_type': None, 'bitstream_path': None, 'data_dim': None,
                   'flag_bitstream_debug': False, 'idx_rate': 0., 'device': 'cuda:0'}
        dev = torch.device(get_value('device', param, default))
        path = get_value('bitstream_path', param, default)
        if not path.endswith(BITSTREAM_SUFFIX):
            path += BITSTREAM_SUFFIX
        with open(path, 'rb') as f:
            frame_bytes = f.read()

        def planes(d):
            return None if d is None else _to_u8_planes(d, dev)
        fc = FrameCodec(self.full_net)
        rec = fc.decode_frame(frame_bytes, planes(get_value('prev_dic', param, default)),
                              planes(get_value('next_dic', param, default)), get_value('frame_type', param, default),
                              get_value('data_dim', param, default), get_value('idx_rate', param, default), dev)
        return _to_float_dic(rec)


def write_yuv(frames, path):
    """frames: list of dicts of uint8 tensors/arrays [1,h,w] -> planar I420 file."""
    with open(path, 'wb') as f:
        for fr in frames:
            for k in 'yuv':
                a = fr[k]
                a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
                f.write(np.ascontiguousarray(a).tobytes())


def decode_one_video(param):
    default = {'decoder': None, 'bitstream_path': '', 'device': 'cuda:0', 'out_file': '', 'flag_bitstream_debug': False}
    decoder = get_value('decoder', param, default).eval()
    path = get_value('bitstream_path', param, default)
    dev = torch.device(get_value('device', param, default))
    out_file = get_value('out_file', param, default)
    print_log_msg('INFO', 'Bitstream path', '', path)
    with open(path, 'rb') as f:
        blob = f.read()
    print_log_msg('INFO', 'Start decoding', '', '')
    t0 = time.time()
    from .. import parallel
    from . import cat_binary_files as container
    rank, world = parallel.rank_world()
    fc = FrameCodec(decoder.full_net)
    with torch.no_grad():
        if world > 1:  # one process per GPU: every rank decodes its intra-period units, rank 0 collects the planes
            _, first, last, _ = container.unpack_video(blob)
            frames = parallel.decode_video_sharded(fc, blob, dev)
        else:
            frames, data_dim, first, last = fc.decode_video(blob, dev)
    torch.cuda.synchronize()
    dt = time.time() - t0
    errs = report_stream_errors(fc, path)  # (every rank reports the sections IT decoded)
    _JOB_ERRORS[0] = len(errs)
    if world > 1:  # ... and the job's total reaches every rank: the exit status of rank 0 speaks for all of them
        import torch.distributed as dist
        cnt = torch.tensor([len(errs)], dtype=torch.int64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(cnt)
        _JOB_ERRORS[0] = int(cnt.item())
    if rank != 0:
        dist_barrier_after_write(world)
        return None
    n = last - first + 1
    print_log_msg('INFO', 'Decoding done', '', '')
    print_log_msg('RESULT', 'Number of frames', '[frame]', int(n))
    print_log_msg('RESULT', 'Decoding time', '[s]', '%.1f' % dt)
    print_log_msg('RESULT', 'Decoding FPS', '[frame/s]', '%.1f' % (n / dt))
    if out_file:
        write_yuv(frames, out_file)
    dist_barrier_after_write(world)
    if get_value('flag_bitstream_debug', param, default):
        check_debug_md5(frames, first, debug_dir(path))
    return frames


STREAM_ERRORS = []  # what the last decode_one_video found on THIS rank
_JOB_ERRORS = [0]   # ... and how many sections failed over all ranks of the job (the CLI's exit status)


def stream_error_count():
    return max(_JOB_ERRORS[0], len(STREAM_ERRORS))


def report_stream_errors(frame_codec, path=''):
    """A bitstream written on another implementation of the transforms (the reference on torch: its sigma differs
    from this build's in the last bits), a damaged file or the wrong model decodes WITHOUT any error from the range
    coder -- it just yields other symbols from the first differing CDF bound on.  Two detectors say so: the md5
    sections of flag_md5sum (src/real_life/bitstream.py:488-499) and, always on, the decoder's bit count against the
    section length.  Printed in the reference's style, not raised; the CLI exits non-zero."""
    errs = frame_codec.stream_errors()
    del STREAM_ERRORS[:]
    STREAM_ERRORS.extend(errs)
    if errs:
        print('-' * 80)
        print('[WARN] %d section(s) of %s did not decode to where their payload ends: the stream was NOT written with the '
              'CDFs this decoder builds (other implementation of the transforms / other model / damaged file); the '
              'frames from the first such section on are unreliable' % (len(errs), path or 'the bitstream'))
        for net, what, i, have, need in errs[:8]:
            print('\t%s %s, stream %d of its launch: payload %d bytes, decode accounts for %d' % (net, what, i, have, need))
        print('-' * 80)
    return errs


def dist_barrier_after_write(world):
    """multi-rank CLI: nobody goes on (to evaluate the output file) before rank 0 has written it"""
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def debug_dir(bitstream_path):
    """where flag_bitstream_debug keeps the encoder's per-plane digests: '<bitstream>.debug/' (the reference uses
    '<root>/debug/<sequence>/' next to its PNG folders, src/model_mngt/model_management.py:132-135)"""
    return bitstream_path + '.debug'


def plane_md5(plane, kind=None):
    """Digest flag_bitstream_debug compares per plane.  The reference hashes the PNG FILE it saved for the plane
    (save_yuv_separately -> to_pil_image(mode='L').save, src/func_util/img_processing.py:290-302; compare at
    src/real_life/decode.py:304-326): the same file is produced here in memory with PIL, so the .md5 files
    interoperate with a reference decoder / encoder running the same Pillow + zlib (the PNG byte stream depends
    on them; pinned by tests/golden/decoder_*.npz `pngmd5_*`).  kind='raw': digest of the raw 8-bit plane,
    independent of Pillow / zlib; kind=None picks 'png' when Pillow is importable."""
    import hashlib
    a = plane.cpu().numpy() if isinstance(plane, torch.Tensor) else np.asarray(plane)
    a = np.ascontiguousarray(a.reshape(a.shape[-2], a.shape[-1]))
    if kind is None:
        kind = digest_kind()
    if kind == 'raw':
        return hashlib.md5(a.tobytes()).hexdigest()
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(a, 'L').save(buf, format='PNG')
    return hashlib.md5(buf.getvalue()).hexdigest()


def digest_kind():
    try:
        import PIL  # noqa: F401
        return 'png'
    except ImportError:
        return 'raw'


def write_debug_md5(frames, first, directory):
    """encoder side of flag_bitstream_debug: '<idx>_<c>.md5' per reconstructed plane (see plane_md5).  A bare
    32-digit digest is the md5 of the PNG file, as in the reference; where Pillow is missing the file says
    'raw:<digest>' so that the decoder compares like with like whatever ITS environment."""
    os.makedirs(directory, exist_ok=True)
    kind = digest_kind()
    for i, fr in enumerate(frames):
        for c in 'yuv':
            with open(os.path.join(directory, '%d_%s.md5' % (first + i, c)), 'w') as f:
                f.write(('raw:' if kind == 'raw' else '') + plane_md5(fr[c], kind))


def check_debug_md5(frames, first, directory):
    """decoder side: same messages as src/real_life/decode.py:318-326; -> number of mismatching planes"""
    bad = 0
    for i, fr in enumerate(frames):
        for c in 'yuv':
            name = '%d_%s' % (first + i, c)
            with open(os.path.join(directory, name + '.md5')) as f:
                encoder_md5 = f.read().strip()
            msg = name + ': '
            kind = 'png'
            if encoder_md5.startswith('raw:'):
                kind, encoder_md5 = 'raw', encoder_md5[4:]
            if encoder_md5 != plane_md5(fr[c], kind):  # (a 'png' digest without Pillow here raises: no silent fallback)
                bad += 1
                msg += '\n' + '-' * 80 + '\n' + 'Incorrect reconstruction!\n' + '-' * 80 + '\n'
            else:
                msg += 'Identical reconstruction!'
            print(msg)
        print('')
    return bad
