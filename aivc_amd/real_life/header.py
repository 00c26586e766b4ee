"""Video / GOP headers of the AIVC container (src/real_life/header.py).

  video header (18 B) = H_x W_x H_y W_y H_z W_z nb_gop idx_first idx_last, uint16 big endian
  GOP header   (6 B)  = flag_LDP(1) nb_chained_gops(2) gop_size(2) round(idx_rate*16)(1)

The byte builders/parsers work in memory; the path-based functions keep the reference's
signatures (they go through a `data_dim.pkl` side file like the reference does).
"""
import math
import os
import pickle

from ..func_util.GOP_structure import generate_gop_struct
from ..func_util.nn_util import get_value
from .utils import GOP_HEADER_SUFFIX, VIDEO_HEADER_SUFFIX

VIDEO_HEADER_SIZE_BYTES = 18
GOP_HEADER_SIZE_BYTES = 6


def video_header_bytes(data_dim, nb_gop, idx_starting_frame, idx_end_frame):
    vals = list(data_dim['x']) + list(data_dim['y']) + list(data_dim['z']) + [nb_gop, idx_starting_frame,
                                                                               idx_end_frame]
    return b''.join(int(v).to_bytes(2, byteorder='big') for v in vals)


def parse_video_header(b):
    v = [int.from_bytes(b[i:i + 2], byteorder='big') for i in range(0, VIDEO_HEADER_SIZE_BYTES, 2)]
    data_dim = {'x': (v[0], v[1]), 'y': (v[2], v[3]), 'z': (v[4], v[5]),
                'x_uv': (math.ceil(v[0] / 2), math.ceil(v[1] / 2))}
    return data_dim, v[6], v[7], v[8]


def gop_header_bytes(GOP_struct_name, idx_rate):
    toks = GOP_struct_name.split('_')
    flag_ldp = 'LDP' in toks
    gop_size = int(toks[-1])
    nb_chained = 0 if flag_ldp else int(toks[0])
    return (int(flag_ldp).to_bytes(1, 'big') + nb_chained.to_bytes(2, 'big') + gop_size.to_bytes(2, 'big')
            + int(round(idx_rate * 16)).to_bytes(1, 'big'))


def parse_gop_header(b):
    flag_ldp = bool(b[0])
    nb_chained = int.from_bytes(b[1:3], 'big')
    gop_size = int.from_bytes(b[3:5], 'big')
    idx_rate = b[5] / 16
    name = 'LDP_%d' % gop_size if flag_ldp else '%d_GOP_%d' % (nb_chained, gop_size)
    return name, idx_rate


def _with_suffix(path, suffix):
    return path if path.endswith(suffix) else path + suffix


def write_video_header(param):
    default = {'header_path': None, 'nb_gop': 0, 'idx_starting_frame': 1, 'idx_end_frame': None}
    path = _with_suffix(get_value('header_path', param, default), VIDEO_HEADER_SUFFIX)
    dd_path = os.path.join(os.path.dirname(path), 'data_dim.pkl')
    with open(dd_path, 'rb') as f:
        data_dim = pickle.load(f)
    os.remove(dd_path)
    with open(path, 'wb') as f:
        f.write(video_header_bytes(data_dim, get_value('nb_gop', param, default),
                                   get_value('idx_starting_frame', param, default),
                                   get_value('idx_end_frame', param, default)))


def read_video_header(param):
    path = _with_suffix(get_value('header_path', param, {'header_path': None}), VIDEO_HEADER_SUFFIX)
    with open(path, 'rb') as f:
        return parse_video_header(f.read())


def write_gop_header(param):
    default = {'header_path': None, 'idx_rate': 0., 'GOP_struct_name': '', 'data_dim': None}
    path = _with_suffix(get_value('header_path', param, default), GOP_HEADER_SUFFIX)
    with open(path, 'wb') as f:
        f.write(gop_header_bytes(get_value('GOP_struct_name', param, default),
                                 get_value('idx_rate', param, default)))
    with open(os.path.join(os.path.dirname(path), 'data_dim.pkl'), 'wb') as f:
        pickle.dump(get_value('data_dim', param, default), f, pickle.HIGHEST_PROTOCOL)


def read_gop_header(param):
    path = _with_suffix(get_value('header_path', param, {'header_path': None}), GOP_HEADER_SUFFIX)
    with open(path, 'rb') as f:
        name, idx_rate = parse_gop_header(f.read())
    return generate_gop_struct(name), idx_rate
