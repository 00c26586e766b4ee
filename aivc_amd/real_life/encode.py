"""encode(param) with the reference's signature and RESULT lines (src/real_life/encode.py), reading
planar YUV directly instead of PNG triplets."""
import os
import time

import numpy as np
import torch

from ..codec import FrameCodec
from ..func_util.console_display import print_log_msg
from ..func_util.nn_util import get_value


def parse_yuv_name(path):
    """'<Name>_<W>x<H>_<fps>_420.yuv' -> (w, h)  (src/format_conversion/utils.py:45-72)"""
    for tok in os.path.basename(path).split('_'):
        if 'x' in tok:
            a, b = tok.split('x')
            if a.isdigit() and b.isdigit():
                return int(a), int(b)
    raise ValueError('cannot parse WxH from %s' % path)


def read_yuv(path, first=0, last=-1, device=None):
    w, h = parse_yuv_name(path)
    hc, wc = (h + 1) // 2, (w + 1) // 2
    fsize = h * w + 2 * hc * wc
    n_total = os.path.getsize(path) // fsize
    last = n_total - 1 if last < 0 else last
    frames = []
    with open(path, 'rb') as f:
        f.seek(first * fsize)
        for _ in range(first, last + 1):
            buf = np.frombuffer(f.read(fsize), np.uint8)
            fr = {'y': buf[:h * w].reshape(1, h, w), 'u': buf[h * w:h * w + hc * wc].reshape(1, hc, wc),
                  'v': buf[h * w + hc * wc:].reshape(1, hc, wc)}
            frames.append({k: torch.from_numpy(v.copy()).to(device) for k, v in fr.items()})
    return frames, first, last


def encode(param):
    default = {'model': None, 'sequence_path': '', 'GOP_struct_name': '', 'GOP_struct': None, 'idx_rate': 0.,
               'final_file': '', 'flag_bitstream_debug': False, 'idx_starting_frame': 0, 'idx_end_frame': -1}
    model = get_value('model', param, default)
    seq = get_value('sequence_path', param, default)
    gop_name = get_value('GOP_struct_name', param, default)
    final_file = get_value('final_file', param, default)
    first = get_value('idx_starting_frame', param, default)
    last = get_value('idx_end_frame', param, default)
    dev = next(model.parameters()).device
    if first > last and last != -1:
        print('ERROR: First frame index bigger than last frame index')
        return
    frames, first, last = read_yuv(seq, first, last, dev)
    print_log_msg('INFO', 'Start encoding', '', '')
    t0 = time.time()
    fc = FrameCodec(model)
    from . import bitstream
    from .. import parallel
    keep_flag, bitstream.ESTIMATE_RATE = bitstream.ESTIMATE_RATE, True  # the reference's in-band rate check (RESULT lines)
    fc.estimated_bits, fc.coded_payload_bytes = 0.0, 0
    rank, world = parallel.rank_world()
    with torch.no_grad():
        if world > 1:  # one process per GPU: intra-period units over the ranks, the container on rank 0
            blob, enc = parallel.encode_video_sharded(fc, frames, gop_name, first, idx_rate=get_value('idx_rate', param, default),
                                                      return_enc=True)
        else:
            enc = fc.encode_video(frames, gop_name, idx_starting_frame=first, idx_end_frame=last,
                                  idx_rate=get_value('idx_rate', param, default))
            blob = fc.assemble_video(enc)
    torch.cuda.synchronize()
    bitstream.ESTIMATE_RATE = keep_flag
    dt = time.time() - t0
    n = last - first + 1
    est_bits, payload = fc.estimated_bits, float(fc.coded_payload_bytes)
    # squared error of the frames THIS process reconstructed (all of them on one GPU), summed over the ranks
    se = cnt = 0.0
    mine = []  # (absolute frame index, reconstruction)
    for u, g in enumerate(enc['recs']):
        if g is None:
            continue
        for i, r in enumerate(g):
            idx = u * len(g) + i
            if idx < n:
                mine.append((first + idx, r))
                se += sum(float(((r[k].float() - frames[idx][k].float()) ** 2).sum()) for k in 'yuv')
                cnt += sum(frames[idx][k].numel() for k in 'yuv')
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([se, cnt, est_bits, payload], dtype=torch.float64, device=parallel._comm_device(None, dev))
        dist.all_reduce(t)
        se, cnt, est_bits, payload = (float(v) for v in t)
    if get_value('flag_bitstream_debug', param, default):
        from .decode import debug_dir, write_debug_md5
        for idx, r in mine:  # every rank writes the digests of its own frames
            write_debug_md5([r], idx, debug_dir(final_file))
    if rank == 0:
        parent = os.path.dirname(final_file)
        if parent:
            os.makedirs(parent, exist_ok=True)
        with open(final_file, 'wb') as f:
            f.write(blob)
    if world > 1:
        dist.barrier()  # the file exists before any rank goes on (to decode it)
    if rank != 0:
        return None
    psnr = 10 * np.log10(255.0 ** 2 / max(se / cnt, 1e-12))
    print_log_msg('INFO', 'Encoding done', '', '')
    print_log_msg('INFO', 'Bitstream path', '', final_file)
    print_log_msg('RESULT', 'Number of frames', '[frame]', int(n))
    print_log_msg('RESULT', 'Encoding/decoding time', '[s]', '%.1f' % dt)
    print_log_msg('RESULT', 'Encoding/decoding FPS', '[frame/s]', '%.1f' % (n / dt))
    print_log_msg('RESULT', 'Estimated PSNR', '[dB]', '%.4f' % psnr)
    # The reference's in-band rate check (src/real_life/encode.py:140-170): the rate its entropy model ESTIMATES against
    # the bytes written.  Here the estimate is what the 16-bit CDF bounds price the coded symbols at (aivc_bounds_rate:
    # sum of -log2((c_hi - c_lo) / 2^16), what an ideal arithmetic coder would write for the same CDFs); the real rate
    # is the file, whose overhead over the estimate is the range coder's flush (< 2 bytes per stream), the map lists
    # and the container's length prefixes and headers.
    est_byte = est_bits / 8
    overhead = (len(blob) / est_byte - 1) * 100 if est_byte > 0 else float('nan')
    print_log_msg('RESULT', 'Estimated rate', '[byte]', '%.1f' % est_byte)
    print_log_msg('RESULT', 'Real rate', '[byte]', len(blob))
    print_log_msg('RESULT', 'Estimated rate overhead', '[%]', '%.2f' % overhead)
    return {'real_rate_byte': len(blob), 'psnr': psnr, 'nb_frames_to_code': n, 'estimated_rate_byte': est_byte,
            'rate_overhead_percent': overhead, 'range_coder_payload_byte': int(payload)}
