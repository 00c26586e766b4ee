"""Container framing (src/real_life/cat_binary_files.py, and the un-cat half of
src/real_life/decode.py:329-426), in memory:

  video = [video header 18 B] ([len 4 B BE][GOP])*
  GOP   = [GOP header 6 B]   ([len 4 B BE][frame])*      frames in DISPLAY order
"""
import glob
import os

from ..func_util.nn_util import get_value
from . import header as hdr
from .utils import BITSTREAM_SUFFIX, GOP_HEADER_SUFFIX, GOP_SUFFIX, VIDEO_HEADER_SUFFIX


def _lp(b):
    return len(b).to_bytes(4, byteorder='big') + b


def pack_gop(gop_header, frames_display_order):
    return gop_header + b''.join(_lp(f) for f in frames_display_order)


def pack_video(video_header, gops):
    return video_header + b''.join(_lp(g) for g in gops)


class ContainerError(ValueError):
    """the byte string is not a complete container: a length prefix points past its end (truncated / damaged file).
    The reference reads the same prefixes with fixed-size file reads and goes on with whatever it got
    (src/real_life/decode.py:329-426); here a short file is an error, not garbage frames."""


def _take(blob, pos, what):
    """the length-prefixed record at pos -> (record, position behind it)"""
    if pos + 4 > len(blob):
        raise ContainerError('%s: length prefix at byte %d lies beyond the end of the data (%d bytes)' % (what, pos, len(blob)))
    n = int.from_bytes(blob[pos:pos + 4], 'big')
    if pos + 4 + n > len(blob):
        raise ContainerError('%s: %d bytes announced at byte %d, only %d left' % (what, n, pos, len(blob) - pos - 4))
    return blob[pos + 4:pos + 4 + n], pos + 4 + n


def unpack_video(blob):
    """-> (data_dim, idx_first, idx_last, [gop bytes])"""
    if len(blob) < hdr.VIDEO_HEADER_SIZE_BYTES:
        raise ContainerError('video header: %d bytes, need %d' % (len(blob), hdr.VIDEO_HEADER_SIZE_BYTES))
    data_dim, nb_gop, first, last = hdr.parse_video_header(blob[:hdr.VIDEO_HEADER_SIZE_BYTES])
    pos, gops = hdr.VIDEO_HEADER_SIZE_BYTES, []
    for g in range(nb_gop):
        rec, pos = _take(blob, pos, 'GOP record %d of %d' % (g, nb_gop))
        gops.append(rec)
    return data_dim, first, last, gops


def unpack_gop(blob):
    """-> (GOP_struct_name, idx_rate, [frame bytes in display order])"""
    if len(blob) < hdr.GOP_HEADER_SIZE_BYTES:
        raise ContainerError('GOP header: %d bytes, need %d' % (len(blob), hdr.GOP_HEADER_SIZE_BYTES))
    name, idx_rate = hdr.parse_gop_header(blob[:hdr.GOP_HEADER_SIZE_BYTES])
    from ..func_util.GOP_structure import generate_gop_struct
    n_frames = len(generate_gop_struct(name))
    pos, frames = hdr.GOP_HEADER_SIZE_BYTES, []
    for i in range(n_frames):
        rec, pos = _take(blob, pos, 'frame %d of %d (%s)' % (i, n_frames, name))
        frames.append(rec)
    return name, idx_rate, frames


# ---- path-based API with the reference's signatures -------------------------------------------
def cat_one_gop(param):
    default = {'idx_gop': 0, 'bitstream_dir': ''}
    idx_gop = get_value('idx_gop', param, default)
    d = get_value('bitstream_dir', param, default)
    d = d if d.endswith('/') else d + '/'
    idxs = []
    for f in glob.glob(d + '*'):
        if f.endswith(GOP_HEADER_SUFFIX) or f.endswith(GOP_SUFFIX) or f.endswith('data_dim.pkl'):
            continue
        idxs.append(int(os.path.basename(f)))
    first = min(idxs)
    gh_path = d + str(idx_gop) + GOP_HEADER_SUFFIX
    with open(gh_path, 'rb') as f:
        gop_header = f.read()
    os.remove(gh_path)
    frames = []
    for i in range(first, first + len(idxs)):
        p = d + str(i) + BITSTREAM_SUFFIX
        with open(p, 'rb') as f:
            frames.append(f.read())
        os.remove(p)
    with open(d + str(idx_gop) + GOP_SUFFIX, 'wb') as f:
        f.write(pack_gop(gop_header, frames))


def cat_one_video(param):
    default = {'bitstream_dir': '', 'idx_starting_frame': 1, 'idx_end_frame': None, 'final_bitstream_path': ''}
    d = get_value('bitstream_dir', param, default)
    d = d if d.endswith('/') else d + '/'
    n_gop = len(glob.glob(d + '*' + GOP_SUFFIX))
    vh = d + VIDEO_HEADER_SUFFIX
    hdr.write_video_header({'nb_gop': n_gop, 'header_path': vh,
                            'idx_starting_frame': get_value('idx_starting_frame', param, default),
                            'idx_end_frame': get_value('idx_end_frame', param, default)})
    with open(vh, 'rb') as f:
        video_header = f.read()
    os.remove(vh)
    gops = []
    for i in range(n_gop):
        p = d + str(i) + GOP_SUFFIX
        with open(p, 'rb') as f:
            gops.append(f.read())
        os.remove(p)
    out = get_value('final_bitstream_path', param, default)
    parent = os.path.dirname(out.rstrip('/'))
    if parent:
        os.makedirs(parent, exist_ok=True)
    with open(out, 'wb') as f:
        f.write(pack_video(video_header, gops))
