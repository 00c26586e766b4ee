"""ctypes mirror of include/aivc_hip.h (struct layouts, constants, prototypes).

The same prototypes are bound twice: on libaivc_hip.so (device pointers, product path) by
aivc_amd/_lib.py, and -- with the ``_ref`` suffix, host pointers -- on the CPU oracle by
oracle/oracle.py (tests only).
"""
import ctypes as C

ABI_VERSION = 17
PREC_FP32, PREC_BF16X3, PREC_FP32_WINO = 0, 1, 2  # aivc_conv_params.precision

AIVC_OK = 0
ERR_UNSUPPORTED = -2
ERRORS = {0: 'AIVC_OK', -1: 'AIVC_ERR_ARG', -2: 'AIVC_ERR_UNSUPPORTED', -3: 'AIVC_ERR_LAUNCH',
          -4: 'AIVC_ERR_WORKSPACE'}

MODE_CONV, MODE_TCONV, MODE_GDN, MODE_IGDN = 0, 1, 2, 3
ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA = 0, 1, 2

AC_MAX_VAL = 256
LP = 514
CDF_ROW = 520
CDF_WIN0, CDF_WIN = 224, 64  # the decoder's fast-path window of a CDF row
BALLE_PARAMS = 43
MAX_MAPS = 256
RC_MAX_STREAMS = 64
RATE_LANES = 16384
WINO_MIN_PIXELS = 8000  # include/aivc_hip.h: AIVC_WINO_MIN_PIXELS
WINO_MIN_PIXELS_TCONV = 32768  # ... AIVC_WINO_MIN_PIXELS_TCONV

FRAME_I, FRAME_P, FRAME_B = 0, 1, 2

_f = C.c_void_p  # every buffer pointer travels as an integer address


CONV_SPARSE4 = 1  # aivc_conv_params.flags: every 4th stored input channel is zero (images padded 3 -> 4)
CONV_WINO_ANY_SIZE = 2  # ... version 2 of the fp32 contract whatever the image size (tests)


class ConvParams(C.Structure):
    _fields_ = [('mode', C.c_int32), ('ksize', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
                ('n', C.c_int32), ('h_in', C.c_int32), ('w_in', C.c_int32), ('c_in', C.c_int32),
                ('h_out', C.c_int32), ('w_out', C.c_int32), ('c_out', C.c_int32),
                ('act1', C.c_int32), ('act2', C.c_int32), ('algo', C.c_int32), ('gdn', C.c_int32),
                ('flags', C.c_int32),
                ('x', _f), ('w', _f), ('bias', _f), ('mul', _f), ('res', _f), ('y', _f),
                ('gdn_beta', _f), ('gdn_gamma', _f),
                ('tail_w', _f), ('tail_bias', _f), ('tail_c_out', C.c_int32), ('precision', C.c_int32),
                ('w_bf16x3', C.c_void_p), ('w_wino', _f)]


MAX_IMAGES = 3


class ImageSrc(C.Structure):
    _fields_ = [('y', _f), ('u', _f), ('v', _f), ('f', _f), ('f_channels', C.c_int32), ('reserved', C.c_int32)]


class MapList(C.Structure):
    _fields_ = [('n_maps', C.c_int32), ('idx', C.c_uint8 * MAX_MAPS)]

    @classmethod
    def make(cls, idx):
        m = cls()
        m.n_maps = len(idx)
        for i, v in enumerate(idx):
            m.idx[i] = int(v)
        return m


def frame_maps_table(maps_list, npix):
    """numpy image of an aivc_frame_maps[len(maps_list)] array (include/aivc_hip.h): frame f codes the channels
    maps_list[f], its positions start where the previous frames' end.  -> (uint8 array [n, 272], [pos_off per frame], total)"""
    import numpy as np
    dt = np.dtype([('pos_off', '<u8'), ('n_maps', '<i4'), ('reserved', '<i4'), ('idx', 'u1', (MAX_MAPS,))])
    t = np.zeros(len(maps_list), dt)
    offs, pos = [], 0
    for f, m in enumerate(maps_list):
        offs.append(pos)
        t['pos_off'][f] = pos
        t['n_maps'][f] = len(m)
        if len(m):
            t['idx'][f, :len(m)] = np.asarray(list(m), np.uint8)
        pos += len(m) * npix
    return t.view(np.uint8).reshape(len(maps_list), dt.itemsize), offs, pos


class RcStream(C.Structure):
    _fields_ = [('in_off', C.c_uint64), ('out_off', C.c_uint64), ('row_off', C.c_uint64),
                ('n_sym', C.c_uint32), ('in_len', C.c_uint32), ('out_cap', C.c_uint32),
                ('plane', C.c_uint32)]


class RcBatch(C.Structure):
    _fields_ = [('n_streams', C.c_int32), ('reserved', C.c_int32), ('s', RcStream * RC_MAX_STREAMS)]


_i32, _sz, _fl = C.c_int32, C.c_size_t, C.c_float
_P = C.POINTER

# name -> argtypes (without the trailing stream argument, which every entry takes)
PROTOTYPES = {
    'aivc_ssim_means': [_f, _f, _i32, _i32, _i32, _f, _i32, C.c_double, C.c_double, _f, _f],
    'aivc_pool2x2': [_f, _i32, _i32, _i32, _i32, _f],
    'aivc_sq_err': [_f, _f, _sz, _f, _f],
    'aivc_conv2d': [_P(ConvParams)],
    'aivc_gdn_reparam': [_f, _f, _i32, _fl, _fl, _fl, _f, _f],
    'aivc_split_weights_bf16x3': [_f, _i32, _i32, C.c_void_p],
    'aivc_winograd_weights': [_f, _i32, _i32, _f],
    'aivc_winograd_weights_poly5': [_f, _i32, _i32, _f],
    'aivc_winograd_weights_tconv5': [_f, _i32, _i32, _f],
    'aivc_pad_channels': [_f, _sz, _i32, _f, _i32],
    'aivc_yuv420_to_444': [_f, _f, _f, _i32, _i32, _i32, _f, _i32, _i32, _i32],
    'aivc_yuv420u8_to_444': [_f, _f, _f, _i32, _i32, _i32, _f, _i32, _i32, _i32],
    'aivc_pack_images': [C.POINTER(ImageSrc), _i32, _i32, _i32, _i32, _f],
    'aivc_conv_images': [C.POINTER(ImageSrc), _i32, C.POINTER(ConvParams)],
    'aivc_frame_to_yuv420': [_f, _i32, _i32, _i32, _i32, _f, _i32, _i32, _i32, _f, _f, _f, _f, _f, _f],
    'aivc_downsample2x': [_f, _i32, _i32, _i32, _i32, _i32, _i32, _f],
    'aivc_warp_blend': [_f, _i32, _i32, _i32, _f, _f, _i32, _i32, _i32, _i32, _i32, _f, _f, _f, _i32, _f, _f],
    'aivc_warp_blend_rows': [_f, _i32, _i32, _i32, _f, _f, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, _f, _f, _i32, _f, _f],
    'aivc_warp': [_f, _f, _i32, _i32, _i32, _i32, _f],
    'aivc_hyper_params': [_f, _i32, _i32, _i32, _i32, _i32, _i32, _f, _f],
    'aivc_channel_gain': [_f, _f, _sz, _i32, _f],
    'aivc_gain_interp': [_f, _f, _i32, _fl, _f],
    'aivc_quantize_center': [_f, _f, _f, _sz, _i32, _f, _f],
    'aivc_dequantize': [_f, _f, _f, _sz, _i32, _f],
    'aivc_balle_cdf_table': [_f, _i32, _f, _f],
    'aivc_nonzero_maps': [_f, _sz, _i32, _f],
    'aivc_nonzero_maps_batch': [_f, _i32, _sz, _i32, _f],
    'aivc_laplace_cdf_rows': [_f, _sz, _i32, _P(MapList), _f],
    'aivc_laplace_cdf_windows': [_f, _sz, _i32, _P(MapList), _f, _f],
    'aivc_laplace_bounds': [_f, _f, _sz, _i32, _P(MapList), _f],
    'aivc_table_bounds': [_f, _f, _sz, _i32, _f],
    'aivc_range_encode': [_f, _P(RcBatch), _f, _f],
    'aivc_range_decode': [_f, _f, _P(RcBatch), _f, _f],
    'aivc_range_decode_windows': [_f, _f, _f, _P(RcBatch), _f, _f],
    'aivc_scatter_symbols': [_f, _sz, _i32, _P(MapList), _f],
    'aivc_laplace_cdf_windows_batch': [_f, _i32, _sz, _i32, _f, _i32, _f, _f],
    'aivc_laplace_bounds_batch': [_f, _f, _i32, _sz, _i32, _f, _i32, _f],
    'aivc_table_bounds_batch': [_f, _f, _i32, _sz, _i32, _f],
    'aivc_scatter_symbols_batch': [_f, _i32, _sz, _i32, _f, _f],
    'aivc_bounds_rate': [_f, _sz, _f, _f],
    'aivc_rate_bits': [_f, _sz, _fl, _fl, _f, _f, _f],
    'aivc_laplace_prob': [_f, _f, _f, _sz, _f],
    'aivc_table_prob': [_f, _f, _sz, _sz, _i32, _f],
}


def declare(lib, suffix=''):
    """Attach argtypes/restype for every entry of the header; raises AttributeError when the
    library does not export one of them."""
    fns = {}
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name + suffix)
        fn.argtypes = list(args) + [C.c_void_p]
        fn.restype = C.c_int
        fns[name] = fn
    if not suffix:
        var = lib.aivc_conv2d_variant
        var.argtypes = [_P(ConvParams)]
        var.restype = C.c_int
        fns['aivc_conv2d_variant'] = var
        chk = lib.aivc_selfcheck_gdn_math  # diagnostic of the device arithmetic: no host twin
        chk.argtypes = [C.c_uint64, C.c_uint32, _f, C.c_void_p]
        chk.restype = C.c_int
        fns['aivc_selfcheck_gdn_math'] = chk
    mws = getattr(lib, 'aivc_metrics_workspace' + suffix)
    mws.argtypes = [_i32, _i32, _i32]
    mws.restype = C.c_size_t
    fns['aivc_metrics_workspace'] = mws
    ver = getattr(lib, 'aivc_abi_version' + suffix)
    ver.argtypes = []
    ver.restype = C.c_int
    fns['aivc_abi_version'] = ver
    return fns


def conv_out_size(mode, h, w, k, stride, pad):
    if mode == MODE_CONV:
        return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    if mode == MODE_TCONV:
        return 2 * h, 2 * w
    return h, w
