"""CPU ORACLE (numpy + oracle/libaivc_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package aivc_amd/ never does.  Feature maps are numpy fp32 arrays in NHWC layout.

Layer semantics restate the reference modules (paths relative to the upstream repo):
  CustomConvLayer / UpscalingLayer / ChengResBlock / ResBlock  src/layers/misc/custom_conv_layers.py
  GDN                                                           src/layers/misc/misc_layers.py:113-154
  AttentionResBlock / SimplifiedAttention                       src/layers/misc/attention.py:22-97
The networks themselves are described by plain-data "specs" (nested dicts holding numpy weights,
see aivc_amd/models/spec.py::export_spec) so the oracle shares no code with the product modules.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from aivc_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libaivc_oracle.so')


def build(force=False):
    """Compile oracle/aivc_oracle.c with the committed Makefile (gcc only, no reference sources)."""
    src = os.path.join(_HERE, 'aivc_oracle.c')
    deps = [src, os.path.join(_HERE, '..', 'include', 'aivc_hip.h'),
            os.path.join(_HERE, '..', 'include', 'aivc_detmath.h')]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return _LIB_PATH
    subprocess.check_call(['make', '-C', _HERE, '-B', 'libaivc_oracle.so'],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_fn = None


def lib():
    global _lib, _fn
    if _lib is None:
        try:
            build()  # no-op when the .so is newer than its sources
        except (OSError, subprocess.CalledProcessError):
            if not os.path.exists(_LIB_PATH):
                raise
        _lib = C.CDLL(_LIB_PATH)
        _fn = abi.declare(_lib, '_ref')
    return _fn


def _p(a):
    return None if a is None else a.ctypes.data


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError('%s_ref failed: %s' % (name, abi.ERRORS.get(rc, rc)))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------------------------
# primitive ops
# ----------------------------------------------------------------------------------------------
def pad_channels(x, c_out):
    x = _f32(x)
    if x.shape[-1] == c_out:
        return x
    out = np.empty(x.shape[:-1] + (c_out,), np.float32)
    _chk(lib()['aivc_pad_channels'](_p(x), x.size // x.shape[-1], x.shape[-1], _p(out), c_out, None),
         'aivc_pad_channels')
    return out


def pack_weight(w_oihw, c_store=None, transposed=False):
    """torch Conv2d weight [O][I][kh][kw] (or ConvTranspose2d [I][O][kh][kw]) -> OHWI, input
    channels zero-padded to c_store."""
    w = np.asarray(w_oihw, np.float32)
    if transposed:
        w = w.transpose(1, 0, 2, 3)
    w = w.transpose(0, 2, 3, 1)
    if c_store is not None and c_store != w.shape[3]:
        w = np.concatenate([w, np.zeros(w.shape[:3] + (c_store - w.shape[3],), np.float32)], axis=3)
    return np.ascontiguousarray(w)


# version of the fp32 contract the conv family computes in (set_precision); the default follows the product's (aivc_amd/ops.py)
PRECISION = {'fp32': abi.PREC_FP32, 'fp32w': abi.PREC_FP32_WINO}[os.environ.get('AIVC_CONTRACT', 'fp32w')]
WINO_ANY_SIZE = False  # tests: version 2 below AIVC_WINO_MIN_PIXELS too (aivc_conv_params.flags)


def set_precision(mode):
    """'fp32' (version 1: the tap chain) / 'fp32w' (version 2, AIVC_PREC_FP32_WINO: Winograd F(2x2, 3x3) chains for the
    stride-1 3x3 layers with c_in % 32 == 0, c_out % 64 == 0); -> the previous mode's name.  The bf16x3 mode has no CPU twin."""
    global PRECISION
    names = {'fp32': abi.PREC_FP32, 'fp32w': abi.PREC_FP32_WINO}
    prev = [k for k, v in names.items() if v == PRECISION][0]
    PRECISION = names[mode]
    return prev


def winograd_weights(w_ohwi, transposed=False):
    """[co, 3, 3, ci] -> co * 16 * ci floats in the staging order of include/aivc_hip.h (AIVC_WINO_U_INDEX); 5x5: the polyphase form
    of a stride-2 kernel (4 ci virtual input channels) or, transposed, its class kernels (4 co virtual output channels)"""
    w_ohwi = _f32(w_ohwi)
    co, k, _, ci = w_ohwi.shape
    assert k in (3, 5) and (k == 5 or not transposed)
    u = np.empty(co * 16 * ci * (4 if k == 5 else 1), np.float32)
    name = 'aivc_winograd_weights_tconv5' if transposed else ('aivc_winograd_weights_poly5' if k == 5 else 'aivc_winograd_weights')
    _chk(lib()[name](_p(w_ohwi), co, ci, _p(u), None), name)
    return u


def conv2d(x, w_ohwi, bias=None, mode=abi.MODE_CONV, stride=1, pad=0, act1=0, act2=0, mul=None,
           res=None, gdn=None, cmap=None, tail=None):
    """gdn = (beta_eff, gamma_eff, inverse): (inverse) GDN fused after the bias.
    tail = (w3 [co2,1,1,co], b3): fused 1x1 tail (include/aivc_hip.h); res / act2 then belong to it, y has co2 channels.
    cmap: stored position of every input channel (default: the first ones, zero padding behind).  The
    accumulation order of the contract is defined on STORED positions (groups of 8 in AIVC_K_ORDER), so an
    input the codec stores as 3-channel images each padded to 4 must be laid out the same way here."""
    x = _f32(x)
    if cmap is not None:
        cmap = list(cmap)
        c_st = (max(cmap) + 4) // 4 * 4
        xs = np.zeros(x.shape[:-1] + (c_st,), np.float32)
        xs[..., cmap] = x
        w_ohwi = _f32(w_ohwi)
        ws = np.zeros(w_ohwi.shape[:3] + (c_st,), np.float32)
        ws[..., cmap] = w_ohwi[..., :len(cmap)]
        x, w_ohwi = xs, ws
    n, h, w_, c = x.shape
    if c % 4:
        x = pad_channels(x, (c + 3) // 4 * 4)
        c = x.shape[-1]
    w_ohwi = _f32(w_ohwi)
    if w_ohwi.shape[3] != c:
        w_ohwi = np.ascontiguousarray(np.concatenate(
            [w_ohwi, np.zeros(w_ohwi.shape[:3] + (c - w_ohwi.shape[3],), np.float32)], axis=3))
    co, k = w_ohwi.shape[0], w_ohwi.shape[1]
    ho, wo = abi.conv_out_size(mode, h, w_, k, stride, pad)
    w3 = b3 = None
    co2 = 0
    if tail is not None:
        w3, b3 = _f32(tail[0]), (None if tail[1] is None else _f32(tail[1]))
        co2 = w3.shape[0]
        assert w3.shape[1:] == (1, 1, co), 'tail weight must be [co2, 1, 1, co]'
    y = np.empty((n, ho, wo, co2 or co), np.float32)
    bias = None if bias is None else _f32(bias)
    mul = None if mul is None else _f32(mul)
    res = None if res is None else _f32(res)
    gb = gg = None
    gflag = 0
    if gdn is not None:
        gb, gg, gflag = _f32(gdn[0]), _f32(gdn[1]), (2 if gdn[2] else 1)
    p = abi.ConvParams(mode, k, stride, pad, n, h, w_, c, ho, wo, co, act1, act2, 0, gflag, abi.CONV_WINO_ANY_SIZE if WINO_ANY_SIZE else 0,
                       _p(x), _p(w_ohwi), _p(bias), _p(mul), _p(res), _p(y), _p(gb), _p(gg), _p(w3), _p(b3), co2, PRECISION)
    _chk(lib()['aivc_conv2d'](C.byref(p), None), 'aivc_conv2d')
    return y


def gdn_reparam(beta, gamma, beta_bound, gamma_bound, pedestal):
    beta, gamma = _f32(beta), _f32(gamma)
    c = beta.shape[0]
    be, ge = np.empty(c, np.float32), np.empty((c, c), np.float32)
    _chk(lib()['aivc_gdn_reparam'](_p(beta), _p(gamma), c, float(beta_bound), float(gamma_bound),
                                   float(pedestal), _p(be), _p(ge), None), 'aivc_gdn_reparam')
    return be, ge


def gdn(x, beta_eff, gamma_eff, inverse=False, res=None):
    c = x.shape[-1]
    mode = abi.MODE_IGDN if inverse else abi.MODE_GDN
    if c % 4 == 0:
        return conv2d(x, _f32(gamma_eff).reshape(c, 1, 1, c), beta_eff, mode=mode, res=res)
    # channel counts that are not a multiple of 4 (never the case in a real model): run on a zero
    # padded copy (extra channels: x = 0, gamma = 0, beta = 1) and drop the padding
    c4 = (c + 3) // 4 * 4
    g = np.zeros((c4, c4), np.float32)
    g[:c, :c] = gamma_eff
    b = np.ones(c4, np.float32)
    b[:c] = beta_eff
    y = conv2d(pad_channels(x, c4), g.reshape(c4, 1, 1, c4), b, mode=mode,
               res=None if res is None else pad_channels(res, c4))
    return np.ascontiguousarray(y[..., :c])


def yuv420_to_444(y, u, v, c_store=3, c_off=0, out=None):
    y, u, v = _f32(y), _f32(u), _f32(v)
    n, h, w = y.shape
    if out is None:
        out = np.zeros((n, h, w, c_store), np.float32)
    _chk(lib()['aivc_yuv420_to_444'](_p(y), _p(u), _p(v), n, h, w, _p(out), out.shape[-1], c_off, 0,
                                     None), 'aivc_yuv420_to_444')
    return out


def yuv420u8_to_444(y, u, v, c_store=3, c_off=0, out=None):
    y, u, v = (np.ascontiguousarray(a, np.uint8) for a in (y, u, v))
    n, h, w = y.shape
    if out is None:
        out = np.zeros((n, h, w, c_store), np.float32)
    _chk(lib()['aivc_yuv420u8_to_444'](_p(y), _p(u), _p(v), n, h, w, _p(out), out.shape[-1], c_off,
                                       0, None), 'aivc_yuv420u8_to_444')
    return out


def pack_images(parts, h, w):
    """oracle twin of ops.pack_images: parts are dicts of uint8 planes [n,h,w], float NHWC arrays or None"""
    n = next(p['y'].shape[0] if isinstance(p, dict) else p.shape[0] for p in parts if p is not None)
    arr = (abi.ImageSrc * abi.MAX_IMAGES)()
    keep = []
    for i, p in enumerate(parts):
        if isinstance(p, dict):
            y, u, v = (np.ascontiguousarray(p[k], np.uint8) for k in 'yuv')
            keep += [y, u, v]
            arr[i].y, arr[i].u, arr[i].v = y.ctypes.data, u.ctypes.data, v.ctypes.data
        elif p is not None:
            f = _f32(p)
            keep.append(f)
            arr[i].f, arr[i].f_channels = f.ctypes.data, f.shape[-1]
    out = np.empty((n, h, w, 4 * len(parts)), np.float32)
    _chk(lib()['aivc_pack_images'](arr, len(parts), n, h, w, _p(out), None), 'aivc_pack_images')
    return out


def conv_images(parts, h, w, w_ohwi, bias=None, act1=0, gdn=None, stride=2, pad=2):
    """oracle twin of aivc_conv_images: conv2d over pack_images(parts) without the caller packing"""
    n = next(p['y'].shape[0] if isinstance(p, dict) else p.shape[0] for p in parts if p is not None)
    arr = (abi.ImageSrc * abi.MAX_IMAGES)()
    keep = []
    for i, p in enumerate(parts):
        if isinstance(p, dict):
            y, u, v = (np.ascontiguousarray(p[k], np.uint8) for k in 'yuv')
            keep += [y, u, v]
            arr[i].y, arr[i].u, arr[i].v = y.ctypes.data, u.ctypes.data, v.ctypes.data
        elif p is not None:
            f = _f32(p)
            keep.append(f)
            arr[i].f, arr[i].f_channels = f.ctypes.data, f.shape[-1]
    w_ohwi = _f32(w_ohwi)
    co, k = w_ohwi.shape[0], w_ohwi.shape[1]
    ho, wo = abi.conv_out_size(abi.MODE_CONV, h, w, k, stride, pad)
    y_out = np.empty((n, ho, wo, co), np.float32)
    bias = None if bias is None else _f32(bias)
    gb = gg = None
    gflag = 0
    if gdn is not None:
        gb, gg, gflag = _f32(gdn[0]), _f32(gdn[1]), (2 if gdn[2] else 1)
    p = abi.ConvParams(abi.MODE_CONV, k, stride, pad, n, h, w, 4 * len(parts), ho, wo, co, act1, 0, 0, gflag, 0,
                       None, _p(w_ohwi), _p(bias), None, None, _p(y_out), _p(gb), _p(gg))
    _chk(lib()['aivc_conv_images'](arr, len(parts), C.byref(p), None), 'aivc_conv_images')
    return y_out


def frame_to_yuv420(x, h, w, skip=None):
    """returns (y, u, v) fp32 8-bit levels and (y8, u8, v8) bytes"""
    x = _f32(x)
    n, hx, wx, cx = x.shape
    skip = None if skip is None else _f32(skip)
    hc, wc = (h + 1) // 2, (w + 1) // 2
    y, u, v = (np.empty((n, h, w), np.float32), np.empty((n, hc, wc), np.float32),
               np.empty((n, hc, wc), np.float32))
    y8, u8, v8 = (np.empty((n, h, w), np.uint8), np.empty((n, hc, wc), np.uint8),
                  np.empty((n, hc, wc), np.uint8))
    _chk(lib()['aivc_frame_to_yuv420'](_p(x), n, hx, wx, cx, _p(skip),
                                       0 if skip is None else skip.shape[-1], h, w, _p(y), _p(u),
                                       _p(v), _p(y8), _p(u8), _p(v8), None), 'aivc_frame_to_yuv420')
    return (y, u, v), (y8, u8, v8)


def downsample2x(x, ch0, nch):
    x = _f32(x)
    n, h, w, c = x.shape
    out = np.empty((n, nch, h // 2, w // 2), np.float32)
    _chk(lib()['aivc_downsample2x'](_p(x), n, h, w, c, ch0, nch, _p(out), None), 'aivc_downsample2x')
    return out


def warp(x, flow):
    x, flow = _f32(x), _f32(flow)
    n, h, w, c = x.shape
    out = np.empty_like(x)
    _chk(lib()['aivc_warp'](_p(x), _p(flow), n, h, w, c, _p(out), None), 'aivc_warp')
    return out


def warp_blend(mof, prev, nxt, h, w, frame_type, co=4, rows=None):
    """rows = (row0, n_rows): the band twin (aivc_warp_blend_rows): mof is the band, prev / nxt whole frames"""
    mof, prev, nxt = _f32(mof), _f32(prev), _f32(nxt)
    n, hm, wm, cm = mof.shape
    row0, nr = (0, h) if rows is None else rows
    pred = np.empty((n, nr, w, co), np.float32)
    skip = np.empty_like(pred)
    xw = np.empty_like(pred)
    alpha = np.empty((n, nr, w), np.float32)
    beta = np.empty((n, nr, w), np.float32)
    _chk(lib()['aivc_warp_blend_rows'](_p(mof), hm, wm, cm, _p(prev), _p(nxt), prev.shape[-1], n, h, w, int(row0), int(nr),
                                       int(frame_type), _p(pred), _p(skip), _p(xw), co, _p(alpha),
                                       _p(beta), None), 'aivc_warp_blend_rows')
    return {'pred': pred, 'skip': skip, 'x_warp': xw, 'alpha': alpha, 'beta': beta}


def hyper_params(hs, c, h, w):
    hs = _f32(hs)
    n, hh, wh, c2 = hs.shape
    assert c2 == 2 * c
    mu = np.empty((n, h, w, c), np.float32)
    sigma = np.empty_like(mu)
    _chk(lib()['aivc_hyper_params'](_p(hs), n, hh, wh, c, h, w, _p(mu), _p(sigma), None),
         'aivc_hyper_params')
    return mu, sigma


def channel_gain(x, gain):
    x = _f32(x)
    gain = None if gain is None else _f32(gain).reshape(-1)
    out = np.empty_like(x)
    _chk(lib()['aivc_channel_gain'](_p(x), _p(gain), x.size // x.shape[-1], x.shape[-1], _p(out),
                                    None), 'aivc_channel_gain')
    return out


def gain_interp(g_r, g_t, lam):
    g_r, g_t = _f32(g_r).reshape(-1), _f32(g_t).reshape(-1)
    out = np.empty_like(g_r)
    _chk(lib()['aivc_gain_interp'](_p(g_r), _p(g_t), g_r.size, float(lam), _p(out), None), 'aivc_gain_interp')
    return out


def quantize_center(y, mu=None, gain_dec=None):
    y = _f32(y)
    mu = None if mu is None else _f32(mu)
    gain_dec = None if gain_dec is None else _f32(gain_dec).reshape(-1)
    q = np.empty(y.shape, np.int16)
    y_hat = np.empty_like(y)
    _chk(lib()['aivc_quantize_center'](_p(y), _p(mu), _p(gain_dec), y.size // y.shape[-1],
                                       y.shape[-1], _p(q), _p(y_hat), None), 'aivc_quantize_center')
    return q, y_hat


def dequantize(q, mu=None, gain_dec=None):
    q = np.ascontiguousarray(q, np.int16)
    mu = None if mu is None else _f32(mu)
    gain_dec = None if gain_dec is None else _f32(gain_dec).reshape(-1)
    out = np.empty(q.shape, np.float32)
    _chk(lib()['aivc_dequantize'](_p(q), _p(mu), _p(gain_dec), q.size // q.shape[-1], q.shape[-1],
                                  _p(out), None), 'aivc_dequantize')
    return out


def pack_balle_params(matrix_h, bias_b, bias_a):
    """lists of numpy arrays as in BallePdfEstim: matrix_h [C,1,3],[C,3,3],[C,3,3],[C,3,1];
    bias_b [C,3]x3,[C,1]; bias_a [C,3]x3  ->  [C][43]"""
    c = matrix_h[0].shape[0]
    parts = [np.asarray(m, np.float32).reshape(c, -1) for m in matrix_h]
    parts += [np.asarray(b, np.float32).reshape(c, -1) for b in bias_b]
    parts += [np.asarray(a, np.float32).reshape(c, -1) for a in bias_a]
    out = np.ascontiguousarray(np.concatenate(parts, axis=1), np.float32)
    assert out.shape[1] == abi.BALLE_PARAMS
    return out


def balle_cdf_table(params):
    params = _f32(params)
    c = params.shape[0]
    table = np.empty((c, abi.CDF_ROW), np.uint16)
    cdf = np.empty((c, abi.LP), np.float32)
    _chk(lib()['aivc_balle_cdf_table'](_p(params), c, _p(table), _p(cdf), None),
         'aivc_balle_cdf_table')
    return table, cdf


def nonzero_maps(q):
    q = np.ascontiguousarray(q, np.int16)
    c = q.shape[-1]
    flags = np.empty(c, np.uint8)
    _chk(lib()['aivc_nonzero_maps'](_p(q), q.size // c, c, _p(flags), None), 'aivc_nonzero_maps')
    return [i for i in range(c) if flags[i]]


def laplace_cdf_rows(sigma, maps):
    sigma = _f32(sigma)
    c = sigma.shape[-1]
    npix = sigma.size // c
    ml = abi.MapList.make(maps)
    rows = np.empty((len(maps) * npix, abi.CDF_ROW), np.uint16)
    _chk(lib()['aivc_laplace_cdf_rows'](_p(sigma), npix, c, C.byref(ml), _p(rows), None),
         'aivc_laplace_cdf_rows')
    return rows


def laplace_tail_mismatches(sigma):
    """entries 0..512 of every sigma: aivc_laplace_cdf_u16_scale (the range decoder's rare path) vs aivc_laplace_cdf_u16
    -> (number of differing entries, (index, k) of the first)"""
    sigma = _f32(sigma).reshape(-1)
    lib()
    fn = _lib.aivc_oracle_laplace_tail_mismatches
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    fi, fk = C.c_int64(-1), C.c_int32(-1)
    bad = fn(_p(sigma), sigma.size, C.byref(fi), C.byref(fk))
    return int(bad), (int(fi.value), int(fk.value))


def laplace_cdf_windows(sigma, maps):
    """-> (win [n_pos][CDF_WIN] uint16, sigma_pos [n_pos] float32): the decoder's fast-path window of every row"""
    sigma = _f32(sigma)
    c = sigma.shape[-1]
    npix = sigma.size // c
    ml = abi.MapList.make(maps)
    win = np.empty((len(maps) * npix, abi.CDF_WIN), np.uint16)
    sp = np.empty(len(maps) * npix, np.float32)
    _chk(lib()['aivc_laplace_cdf_windows'](_p(sigma), npix, c, C.byref(ml), _p(win), _p(sp), None),
         'aivc_laplace_cdf_windows')
    return win, sp


def range_decode_windows(payload, win, sigma_pos, n_sym, want_bits=False):
    win = np.ascontiguousarray(win, np.uint16)
    sigma_pos = _f32(sigma_pos)
    buf = np.frombuffer(payload, np.uint8)
    padded = np.zeros((len(buf) + 3) // 4 * 4 + 8, np.uint8)
    padded[:len(buf)] = buf
    sym = np.empty(n_sym, np.uint16)
    b = abi.RcBatch()
    b.n_streams = 1
    s = b.s[0]
    s.in_off, s.out_off, s.row_off, s.n_sym, s.in_len, s.plane = 0, 0, 0, n_sym, len(buf), 0
    bits = np.zeros(1, np.uint32)
    _chk(lib()['aivc_range_decode_windows'](_p(padded), _p(win), _p(sigma_pos), C.byref(b), _p(sym), _p(bits), None),
         'aivc_range_decode_windows')
    return (sym, int(bits[0])) if want_bits else sym


def laplace_bounds(sigma, q, maps):
    sigma = _f32(sigma)
    q = np.ascontiguousarray(q, np.int16)
    c = sigma.shape[-1]
    npix = sigma.size // c
    ml = abi.MapList.make(maps)
    bounds = np.empty(len(maps) * npix, np.uint32)
    _chk(lib()['aivc_laplace_bounds'](_p(sigma), _p(q), npix, c, C.byref(ml), _p(bounds), None),
         'aivc_laplace_bounds')
    return bounds


def table_bounds(table, q):
    table = np.ascontiguousarray(table, np.uint16)
    q = np.ascontiguousarray(q, np.int16)
    c = q.shape[-1]
    npix = q.size // c
    bounds = np.empty(c * npix, np.uint32)
    _chk(lib()['aivc_table_bounds'](_p(table), _p(q), npix, c, _p(bounds), None),
         'aivc_table_bounds')
    return bounds


def bounds_rate(bounds):
    """bits the range coder pays for these packed CDF bounds (sum of -log2 of the coded probabilities, fp64)"""
    bounds = np.ascontiguousarray(bounds, np.uint32)
    lanes, out = np.zeros(abi.RATE_LANES, np.float64), np.zeros(1, np.float64)
    _chk(lib()['aivc_bounds_rate'](_p(bounds) if bounds.size else None, bounds.size, _p(lanes), _p(out), None), 'aivc_bounds_rate')
    return float(out[0])


def rate_bits(prob, p_min, p_max):
    """EntropyCoder.forward -> (rate fp32 like prob, fp64 sum)"""
    prob = np.ascontiguousarray(prob, np.float32)
    rate = np.empty_like(prob)
    lanes, out = np.zeros(abi.RATE_LANES, np.float64), np.zeros(1, np.float64)
    _chk(lib()['aivc_rate_bits'](_p(prob), prob.size, p_min, p_max, _p(rate), _p(lanes), _p(out), None), 'aivc_rate_bits')
    return rate, float(out[0])


def laplace_prob(y, mu, sigma):
    y = np.ascontiguousarray(y, np.float32)
    sigma = np.ascontiguousarray(sigma, np.float32)
    mu = None if mu is None else np.ascontiguousarray(mu, np.float32)
    prob = np.empty_like(y)
    _chk(lib()['aivc_laplace_prob'](_p(y), None if mu is None else _p(mu), _p(sigma), y.size, _p(prob), None), 'aivc_laplace_prob')
    return prob


def table_prob(x, cdf_f32):
    """x [b, c, h, w] integer-valued floats, cdf_f32 [c, 514] -> probabilities [b, c, h, w]"""
    x = np.ascontiguousarray(x, np.float32)
    cdf_f32 = np.ascontiguousarray(cdf_f32, np.float32)
    prob = np.empty_like(x)
    _chk(lib()['aivc_table_prob'](_p(x), _p(cdf_f32), x.size, x.shape[2] * x.shape[3], x.shape[1], _p(prob), None), 'aivc_table_prob')
    return prob


def range_encode(bounds):
    """one stream -> bytes"""
    bounds = np.ascontiguousarray(bounds, np.uint32)
    cap = 16 + bounds.size * 3
    out = np.zeros(cap, np.uint8)
    out_len = np.zeros(1, np.uint32)
    b = abi.RcBatch()
    b.n_streams = 1
    b.s[0].in_off, b.s[0].out_off, b.s[0].n_sym, b.s[0].out_cap = 0, 0, bounds.size, cap
    _chk(lib()['aivc_range_encode'](_p(bounds), C.byref(b), _p(out), _p(out_len), None),
         'aivc_range_encode')
    assert out_len[0] != 0xFFFFFFFF
    return out[:out_len[0]].tobytes()


def range_decode(payload, rows, n_sym, plane=0, want_bits=False):
    """rows: [n_rows][CDF_ROW] uint16; plane=0 -> one row per symbol, else row = i // plane.
    want_bits: -> (sym, bits consumed); an intact stream has len(payload) == (bits + 2 + 7) // 8 (include/aivc_hip.h)"""
    rows = np.ascontiguousarray(rows, np.uint16)
    buf = np.frombuffer(payload, np.uint8)
    padded = np.zeros((len(buf) + 3) // 4 * 4 + 8, np.uint8)
    padded[:len(buf)] = buf
    sym = np.empty(n_sym, np.uint16)
    b = abi.RcBatch()
    b.n_streams = 1
    s = b.s[0]
    s.in_off, s.out_off, s.row_off, s.n_sym, s.in_len, s.plane = 0, 0, 0, n_sym, len(buf), plane
    bits = np.zeros(1, np.uint32)
    _chk(lib()['aivc_range_decode'](_p(padded), _p(rows), C.byref(b), _p(sym), _p(bits), None),
         'aivc_range_decode')
    return (sym, int(bits[0])) if want_bits else sym


def scatter_symbols(sym, npix, c, maps):
    sym = np.ascontiguousarray(sym, np.uint16)
    q = np.empty((npix, c), np.int16)
    ml = abi.MapList.make(maps)
    _chk(lib()['aivc_scatter_symbols'](_p(sym), npix, c, C.byref(ml), _p(q), None),
         'aivc_scatter_symbols')
    return q


# ----------------------------------------------------------------------------------------------
# layers (specs are nested dicts; x is NHWC)
# ----------------------------------------------------------------------------------------------
_ACT = {'no': abi.ACT_NONE, None: abi.ACT_NONE, 'leaky_relu': abi.ACT_LEAKY, 'relu': abi.ACT_RELU,
        'sigmoid': abi.ACT_SIGMOID}


def _gdn_from_spec(g, x, res=None):
    be, ge = gdn_reparam(g['beta'], g['gamma'], g['beta_bound'], g['gamma_bound'], g['pedestal'])
    return gdn(x, be, ge, inverse=g['inverse'], res=res)


def image_cmap(n_images):
    """stored channel positions of n 3-channel images, each padded to 4 (the codec's layout of its image
    inputs: aivc_amd/codec.py FrameCodec._images)"""
    return tuple(4 * i + c for i in range(n_images) for c in range(3))


def run_layer(spec, x, res=None, cmap=None):
    """Evaluate one layer spec.  `res` (optional) is added to the output (used by the residual
    blocks to express `aux(x) + layers(x)` with the same operand order as the fused kernels).
    cmap: stored layout of the input channels, for the first conv of a transform (see conv2d)."""
    t = spec['type']
    if t == 'Sequential':
        if len(spec['layers']) == 1:
            return run_layer(spec['layers'][0], x, res=res, cmap=cmap)
        x = run_layer(spec['layers'][0], x, cmap=cmap)
        for s in spec['layers'][1:-1]:
            x = run_layer(s, x)
        return run_layer(spec['layers'][-1], x, res=res)
    if t == 'CustomConvLayer':  # custom_conv_layers.py:129-180
        k = spec['k']
        w = pack_weight(spec['weight'])
        nl = spec['nl']
        if nl in ('gdn', 'gdn_inverse'):
            y = conv2d(x, w, spec.get('bias'), stride=spec['stride'], pad=k // 2, cmap=cmap)
            return _gdn_from_spec(spec['gdn'], y, res=res)
        return conv2d(x, w, spec.get('bias'), stride=spec['stride'], pad=k // 2, act1=_ACT[nl],
                      res=res, cmap=cmap)
    if t == 'UpscalingLayer':  # custom_conv_layers.py:183-253
        w = pack_weight(spec['weight'], transposed=True)
        nl = spec['nl']
        if nl in ('gdn', 'gdn_inverse'):
            y = conv2d(x, w, spec.get('bias'), mode=abi.MODE_TCONV, stride=2)
            return _gdn_from_spec(spec['gdn'], y, res=res)
        return conv2d(x, w, spec.get('bias'), mode=abi.MODE_TCONV, stride=2, act1=_ACT[nl], res=res)
    if t == 'Conv2d':  # bare nn.Conv2d (1x1 in the attention blocks / the s2 skip of ChengResBlock)
        return conv2d(x, pack_weight(spec['weight']), spec.get('bias'), stride=spec['stride'], pad=0,
                      act1=_ACT[spec.get('nl')], res=res, mul=spec.get('_mul'))
    if t == 'ChengResBlock':  # custom_conv_layers.py:21-109
        if spec['mode'] == 'plain':
            return run_layer(spec['layers'], x, res=x)  # x + layers(x)
        aux = run_layer(spec['aux'], x)
        return run_layer(spec['layers'], x, res=aux)  # aux(x) + layers(x)
    if t == 'ResBlock':  # custom_conv_layers.py:112-126 : relu(x + conv(relu(conv(x))))
        k = spec['k']
        h = conv2d(x, pack_weight(spec['w1']), spec['b1'], pad=k // 2, act1=abi.ACT_RELU)
        return conv2d(h, pack_weight(spec['w2']), spec['b2'], pad=k // 2, res=x, act2=abi.ACT_RELU)
    if t == 'AttentionResBlock':  # attention.py:22-42 : leaky(x + c1x1(leaky(c3x3(leaky(c1x1(x))))))
        h = conv2d(x, pack_weight(spec['w1']), spec['b1'], act1=abi.ACT_LEAKY)
        h = conv2d(h, pack_weight(spec['w2']), spec['b2'], pad=1, act1=abi.ACT_LEAKY)
        return conv2d(h, pack_weight(spec['w3']), spec['b3'], res=x, act2=abi.ACT_LEAKY)
    if t == 'SimplifiedAttention':  # attention.py:45-97 : trunk(x) * sigmoid(conv1x1(att(x))) + x
        trunk = x
        for s in spec['trunk']:
            trunk = run_layer(s, trunk)
        att = x
        for s in spec['attention']:
            att = run_layer(s, att)
        return conv2d(att, pack_weight(spec['w_out']), spec['b_out'], act1=abi.ACT_SIGMOID,
                      mul=trunk, res=x)
    raise ValueError('unknown layer spec type %r' % t)


# ---- quality metrics ---------------------------------------------------------------------------------
def _f64(x):
    return np.ascontiguousarray(np.asarray(x, np.float64))


def ssim_means(a, b, win, c1, c2):
    a, b, win = _f64(a), _f64(b), _f64(win)
    n, h, w = a.shape
    out = np.empty((n, 2), np.float64)
    ws = np.empty(1024, np.float64)
    _chk(lib()['aivc_ssim_means'](_p(a), _p(b), n, h, w, _p(win), len(win), float(c1), float(c2), _p(ws), _p(out), None),
         'aivc_ssim_means')
    return out


def pool2x2(x, edge):
    x = _f64(x)
    n, h, w = x.shape
    out = np.empty((n, (h + 1) // 2, (w + 1) // 2), np.float64)
    _chk(lib()['aivc_pool2x2'](_p(x), n, h, w, int(edge), _p(out), None), 'aivc_pool2x2')
    return out


def sq_err(a, b):
    a, b = _f64(a), _f64(b)
    out = np.empty(1, np.float64)
    ws = np.empty(1024, np.float64)
    _chk(lib()['aivc_sq_err'](_p(a), _p(b), a.size, _p(ws), _p(out), None), 'aivc_sq_err')
    return out
