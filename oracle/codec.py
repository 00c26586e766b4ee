"""CPU ORACLE of the frame / GOP / video codec (numpy orchestration over oracle.py).  TEST
INFRASTRUCTURE ONLY.

decode_frame restates Decoder.decode / MOFNetDecoder / CodecNetDecoder / ConditionalDecoder
(src/real_life/decode.py:455-898); encode_frame is its mirror image (the reference's
FullNet.GOP_forward is absent from the snapshot); the container follows
src/real_life/header.py:25-40 and src/real_life/cat_binary_files.py:19-198; GOP structures follow
src/func_util/GOP_structure.py:27-221.  Frames are dicts of uint8 numpy planes [h,w].
"""
import math

import numpy as np

from . import oracle as O

FRAME_I, FRAME_P, FRAME_B = 0, 1, 2
SECTIONS = ('mofnet_z', 'mofnet_y', 'codecnet_z', 'codecnet_y')


# ---- GOP structures (src/func_util/GOP_structure.py) --------------------------------------------
def gop_struct(name):
    """-> {display idx: (type, prev, next, coding_order)}"""
    toks = name.split('_')
    if name == '1_GOP_0':
        return {0: (FRAME_I, None, None, 0)}
    if 'LDP' in toks:
        n = int(toks[-1])
        g = {0: (FRAME_I, None, None, 0)}
        for i in range(1, n + 1):
            g[i] = (FRAME_P, i - 1, None, i)
        return g
    size, chain = int(toks[-1]), int(toks[0])
    g = {0: (FRAME_I, None, None, 0)}
    for c in range(chain):
        off = c * size
        g[off + size] = (FRAME_P, off, None, off + 1)
        cnt = [off + 2]

        def rec(mid, half):
            g[mid] = (FRAME_B, mid - half, mid + half, cnt[0])
            cnt[0] += 1
            half //= 2
            if half:
                rec(mid - half, half)
                rec(mid + half, half)
        rec(off + size // 2, size // 2)
    return g


# ---- container ----------------------------------------------------------------------------------
def be(v, n):
    return int(v).to_bytes(n, 'big')


def video_header(data_dim, nb_gop, first, last):
    return b''.join(be(v, 2) for v in (*data_dim['x'], *data_dim['y'], *data_dim['z'], nb_gop, first, last))


def gop_header(name, idx_rate):
    toks = name.split('_')
    ldp = 'LDP' in toks
    return be(int(ldp), 1) + be(0 if ldp else int(toks[0]), 2) + be(int(toks[-1]), 2) + be(int(round(idx_rate * 16)), 1)


def lp(b):
    return be(len(b), 4) + b


def split_lp(blob, pos, count):
    out = []
    for _ in range(count):
        n = int.from_bytes(blob[pos:pos + 4], 'big')
        out.append(blob[pos + 4:pos + 4 + n])
        pos += 4 + n
    return out


# ---- conditional coder ----------------------------------------------------------------------------
def _gain(net, frame_type, mode, idx_rate=0):
    key = 'I' if (not net['flag_gain_p_b'] or frame_type == FRAME_I) else ('P' if frame_type == FRAME_P else 'B')
    lst = net['gain'][key][mode]
    if float(idx_rate) == int(idx_rate):
        return lst[int(idx_rate)]
    prev_i = int(math.floor(idx_rate))
    next_i = prev_i + 1 if prev_i + 1 < len(lst) else prev_i
    return O.gain_interp(lst[prev_i], lst[next_i], 1 - (idx_rate - prev_i))


def _shortcut(net, in_shortcut, h_y, w_y):
    if in_shortcut is not None and net['g_a_ref'] is not None:
        return O.run_layer(net['g_a_ref'], in_shortcut, cmap=O.image_cmap(in_shortcut.shape[-1] // 3))[:, :h_y, :w_y, :]
    return np.zeros((1, h_y, w_y, net['c_short']), np.float32)


def _z_section(table, q_z):
    return O.range_encode(O.table_bounds(table, q_z))


def _y_section(sigma, q_y):
    maps = O.nonzero_maps(q_y)
    body = bytes([len(maps)]) + bytes(maps)
    if maps:
        body += O.range_encode(O.laplace_bounds(sigma, q_y, maps))
    return body


def cond_encode(net, x_in, in_shortcut, frame_type, idx_rate=0.):
    y = O.run_layer(net['g_a'], x_in, cmap=O.image_cmap(x_in.shape[-1] // 3))  # images: 3 channels stored as 4
    y = O.channel_gain(y, _gain(net, frame_type, 'enc', idx_rate))
    z = O.run_layer(net['h_a'], y)
    q_z, z_hat = O.quantize_center(z)
    h_y, w_y = y.shape[1:3]
    mu, sigma = O.hyper_params(O.run_layer(net['h_s'], z_hat), net['c_y'], h_y, w_y)
    q_y, y_hat = O.quantize_center(y, mu, _gain(net, frame_type, 'dec', idx_rate))
    table, _ = O.balle_cdf_table(net['balle'])
    s = _shortcut(net, in_shortcut, h_y, w_y)
    x_out = O.run_layer(net['g_s'], np.concatenate((y_hat, s), axis=3))
    return x_out, _z_section(table, q_z), _y_section(sigma, q_y), (h_y, w_y), tuple(z.shape[1:3])


def cond_decode(net, sec_z, sec_y, in_shortcut, frame_type, dim_y, dim_z, idx_rate=0., sigma_hook=None):
    """src/real_life/decode.py:798-898
    sigma_hook(sigma NHWC) -> sigma: test aid for streams written by ANOTHER implementation of the transforms (the
    reference on torch): its h_s rounds differently in the last bits, one differing CDF count on a coded symbol
    desynchronises any arithmetic decoder, so such tests hand the writer's sigma to the CDF build (and check the own
    sigma against it separately); everything else stays this decoder's own dataflow."""
    table, _ = O.balle_cdf_table(net['balle'])
    c_z, c_y = net['c_z'], net['c_y']
    npz = dim_z[0] * dim_z[1]
    sym = O.range_decode(sec_z, table, c_z * npz, plane=npz)
    q_z = O.scatter_symbols(sym, npz, c_z, list(range(c_z))).reshape(1, dim_z[0], dim_z[1], c_z)
    z_hat = O.dequantize(q_z)
    mu, sigma = O.hyper_params(O.run_layer(net['h_s'], z_hat), c_y, dim_y[0], dim_y[1])
    if sigma_hook is not None:
        sigma = np.ascontiguousarray(sigma_hook(sigma), np.float32)
    npy = dim_y[0] * dim_y[1]
    n_maps = sec_y[0]
    maps = list(sec_y[1:1 + n_maps])
    if n_maps:
        sym = O.range_decode(sec_y[1 + n_maps:], O.laplace_cdf_rows(sigma, maps), n_maps * npy)
        q_y = O.scatter_symbols(sym, npy, c_y, maps).reshape(1, dim_y[0], dim_y[1], c_y)
    else:
        q_y = np.zeros((1, dim_y[0], dim_y[1], c_y), np.int16)
    y_hat = O.dequantize(q_y, mu, _gain(net, frame_type, 'dec', idx_rate))
    s = _shortcut(net, in_shortcut, dim_y[0], dim_y[1])
    return O.run_layer(net['g_s'], np.concatenate((y_hat, s), axis=3))


# ---- frame ---------------------------------------------------------------------------------------
def to444(planes, h, w):
    if planes is None:
        return np.zeros((1, h, w, 3), np.float32)
    return O.yuv420u8_to_444(planes['y'][None], planes['u'][None], planes['v'][None], c_store=3)


def _rec(cod_out, h, w, skip):
    _, (y8, u8, v8) = O.frame_to_yuv420(cod_out, h, w, skip=skip)
    return {'y': y8[0], 'u': u8[0], 'v': v8[0]}


def encode_frame(model, cur, prev, nxt, frame_type, idx_rate=0.):
    h, w = cur['y'].shape
    code = to444(cur, h, w)
    secs = [b'', b'', None, None]
    empty = [True, True, False, False]
    pred = skip = None
    if frame_type != FRAME_I:
        p444, n444 = to444(prev, h, w), to444(nxt if frame_type == FRAME_B else None, h, w)
        short_in = np.concatenate((p444, n444), axis=3) if frame_type == FRAME_B else None
        mof_out, secs[0], secs[1], _, _ = cond_encode(model['mof'], np.concatenate((code, p444, n444), axis=3),
                                                      short_in, frame_type, idx_rate)
        empty[0] = empty[1] = False
        wb = O.warp_blend(mof_out, p444, n444, h, w, frame_type, co=3)
        pred, skip = wb['pred'], wb['skip']
    zero = np.zeros_like(code) if pred is None else pred
    cod_out, secs[2], secs[3], dim_y, dim_z = cond_encode(model['cod'], np.concatenate((code, zero), axis=3), pred,
                                                          frame_type, idx_rate)
    frame = b''.join(be(0, 4) if empty[i] else lp(secs[i]) for i in range(4))
    data_dim = {'x': (h, w), 'y': dim_y, 'z': dim_z}
    return frame, _rec(cod_out, h, w, skip), data_dim


def decode_frame(model, frame_bytes, prev, nxt, frame_type, data_dim, idx_rate=0., sigma_hook=None):
    """src/real_life/decode.py:455-580; sigma_hook(net name 'mofnet' / 'codecnet', sigma) -> sigma, see cond_decode"""
    hook = (lambda name: (lambda s: sigma_hook(name, s))) if sigma_hook is not None else (lambda name: None)
    h, w = data_dim['x']
    sec = split_lp(frame_bytes, 0, 4)
    pred = skip = None
    if frame_type != FRAME_I:
        p444, n444 = to444(prev, h, w), to444(nxt if frame_type == FRAME_B else None, h, w)
        short_in = np.concatenate((p444, n444), axis=3) if frame_type == FRAME_B else None
        mof_out = cond_decode(model['mof'], sec[0], sec[1], short_in, frame_type, data_dim['y'], data_dim['z'], idx_rate,
                              hook('mofnet'))
        wb = O.warp_blend(mof_out, p444, n444, h, w, frame_type, co=3)
        pred, skip = wb['pred'], wb['skip']
    cod_out = cond_decode(model['cod'], sec[2], sec[3], pred, frame_type, data_dim['y'], data_dim['z'], idx_rate,
                          hook('codecnet'))
    return _rec(cod_out, h, w, skip)


# ---- GOP / video ------------------------------------------------------------------------------------
def encode_video(model, frames, gop_name, first=0, idx_rate=0.):
    n = len(frames)
    g = gop_struct(gop_name)
    unit = len(g)
    nb_gop = math.ceil(n / unit)
    order = sorted(g, key=lambda i: g[i][3])
    gops, recs, data_dim = [], [], None
    for u in range(nb_gop):
        chunk = [frames[min(u * unit + i, n - 1)] for i in range(unit)]
        rec, fb = {}, {}
        for i in order:
            t, p, nx, _ = g[i]
            fb[i], rec[i], data_dim = encode_frame(model, chunk[i], rec.get(p), rec.get(nx), t, idx_rate)
        gops.append(gop_header(gop_name, idx_rate) + b''.join(lp(fb[i]) for i in sorted(g)))
        recs.extend(rec[i] for i in sorted(g))
    blob = video_header(data_dim, nb_gop, first, first + n - 1) + b''.join(lp(x) for x in gops)
    return blob, recs[:n]


def decode_video(model, blob, sigma_hook=None):
    """sigma_hook(absolute frame index, net name, sigma) -> sigma (test aid, see cond_decode)"""
    v = [int.from_bytes(blob[i:i + 2], 'big') for i in range(0, 18, 2)]
    data_dim = {'x': (v[0], v[1]), 'y': (v[2], v[3]), 'z': (v[4], v[5])}
    nb_gop, first, last = v[6], v[7], v[8]
    out = []
    for gi, gb in enumerate(split_lp(blob, 18, nb_gop)):
        ldp, chain, size = bool(gb[0]), int.from_bytes(gb[1:3], 'big'), int.from_bytes(gb[3:5], 'big')
        name = 'LDP_%d' % size if ldp else '%d_GOP_%d' % (chain, size)
        idx_rate = gb[5] / 16
        g = gop_struct(name)
        fbytes = split_lp(gb, 6, len(g))
        rec = {}
        for i in sorted(g, key=lambda i: g[i][3]):
            t, p, nx, _ = g[i]
            fh = None if sigma_hook is None else (lambda name, s, idx=first + gi * len(g) + i: sigma_hook(idx, name, s))
            rec[i] = decode_frame(model, fbytes[i], rec.get(p), rec.get(nx), t, data_dim, idx_rate, fh)
        out.extend(rec[i] for i in sorted(g))
    return out[:last - first + 1]
