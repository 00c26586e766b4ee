"""Tensor-level wrappers over the C ABI (include/aivc_hip.h).

PyTorch is used for device memory and streams only: every function takes CUDA tensors, hands raw
device pointers to libaivc_hip.so on torch's current stream and returns freshly allocated CUDA
tensors.  Feature maps are NHWC fp32 ([n, h, w, c], contiguous).  CPU tensors are rejected: the
product has no CPU execution path (the CPU restatement lives in oracle/ and is test-only).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import abi
from ._lib import AivcNativeError, call, load

FRAME_I, FRAME_P, FRAME_B = abi.FRAME_I, abi.FRAME_P, abi.FRAME_B

# when bench.py sets this to a list, every aivc_conv2d launch is bracketed by HIP events
PROFILE = None
PROFILE_DIRECT_EQUIVALENT = [0.0, 0.0]  # Winograd launches while PROFILE is on: [tap-chain FLOPs they replace, FLOPs they execute]
# likewise for the HBM-bound stages: (name, algorithmic bytes, event, event) per launch
PROFILE_HBM = None


def _hbm_profiled(name, nbytes, launch):
    if PROFILE_HBM is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    PROFILE_HBM.append((name, float(nbytes), e0, e1))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    # hipStream_t of torch's current stream on the current device (the raw getter avoids ~7 us of
    # Python per launch)
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype=None, name='tensor'):
    if t is None:
        return None
    if isinstance(t, ImageStack):  # used as anything but the input of the first conv: materialise
        t = t.tensor()
    if not t.is_cuda:
        raise AivcNativeError('aivc_amd.ops: %s is on %s; the HIP path needs CUDA tensors '
                              '(no CPU fallback)' % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise AivcNativeError('aivc_amd.ops: %s has dtype %s, expected %s' % (name, t.dtype, dtype))
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


# NCHW (logical, torch module API) <-> NHWC (physical, kernels) -------------------------------
def to_nhwc(x):
    """[n,c,h,w] tensor (any strides) -> contiguous [n,h,w,c] (free when x is channels_last)."""
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw_view(x):
    """[n,h,w,c] contiguous -> logical [n,c,h,w] view (channels_last strides, no copy)."""
    return x.permute(0, 3, 1, 2)


def pad_channels(x, c_out):
    x = _dev(x, torch.float32, 'x')
    c_in = x.shape[-1]
    if c_in == c_out:
        return x
    out = torch.empty(x.shape[:-1] + (c_out,), dtype=torch.float32, device=x.device)
    call('aivc_pad_channels', _p(x), x.numel() // c_in, c_in, _p(out), c_out, _stream())
    return out


def pack_weight(w, c_store=None, transposed=False):
    """torch Conv2d weight [O,I,kh,kw] (ConvTranspose2d: [I,O,kh,kw]) -> OHWI fp32 contiguous with
    the input-channel axis zero padded to c_store (pure data movement, done once per layer)."""
    w = w.detach()
    if transposed:
        w = w.permute(1, 0, 2, 3)
    w = w.permute(0, 2, 3, 1)
    ci = w.shape[3]
    c_store = ci if c_store is None else c_store
    if c_store != ci:
        wp = torch.zeros(w.shape[:3] + (c_store,), dtype=torch.float32, device=w.device)
        wp[..., :ci] = w
        return wp
    return w.contiguous().float()


# The arithmetic contract the conv family computes in (see set_precision()).  Default since round 6: VERSION 2 of the fp32
# contract ('fp32w': Winograd F(2x2, 3x3) chains for the stride-1 3x3 layers it covers) -- as exact as version 1 in the sense
# that matters (HIP == CPU oracle bit for bit, bitstreams reproducible on every GPU), closer to the reference's outputs on the
# layer fixtures, and 1.4 ... 1.6x faster on the layers it covers.  AIVC_CONTRACT=fp32 in the environment selects version 1
# (an encoder and a decoder must run the same version: their bits differ).
_CONTRACT_NAMES = {'fp32': abi.PREC_FP32, 'fp32w': abi.PREC_FP32_WINO}
DEFAULT_CONTRACT = os.environ.get('AIVC_CONTRACT', 'fp32w')
if DEFAULT_CONTRACT not in _CONTRACT_NAMES:
    raise ValueError('AIVC_CONTRACT=%r: expected fp32 or fp32w' % DEFAULT_CONTRACT)
PRECISION = _CONTRACT_NAMES[DEFAULT_CONTRACT]


def set_precision(mode):
    """'fp32': version 1 of the arithmetic contract (9-tap chains) -- HIP == CPU oracle bit for bit, bitstreams reproducible on every GPU.
    'bf16x3': the precision MODE of the wide convolutions (include/aivc_hip.h, aivc_conv_params.precision): fp32 operands
    as three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulation.  Results are within fp32
    summation-order noise of the contract's, not its bits: an encoder and a decoder must run the same mode.
    'fp32w' (the default, DEFAULT_CONTRACT): version 2 of the fp32 contract (AIVC_PREC_FP32_WINO): the stride-1 3x3 layers with c_in % 32 == 0 and
    c_out % 64 == 0 accumulate along Winograd F(2x2, 3x3) chains (16 instead of 36 multiplications per 2x2 outputs);
    everything else as 'fp32'.  HIP == CPU oracle bit for bit in this version too; its bits are not version 1's.
    -> the previous mode's name.  Process-wide (the codec's side streams read it too)."""
    global PRECISION
    names = {'fp32': abi.PREC_FP32, 'bf16x3': abi.PREC_BF16X3, 'fp32w': abi.PREC_FP32_WINO}
    if mode not in names:
        raise ValueError('precision %r: expected one of %s' % (mode, sorted(names)))
    prev = [k for k, v in names.items() if v == PRECISION][0]
    PRECISION = names[mode]
    return prev


_SPLIT_WEIGHTS = {}  # id(weight tensor) -> (weak reference, (data_ptr, version), its bf16x3 image); dies with the packed weight


def split_weights_bf16x3(w_ohwi):
    """aivc_split_weights_bf16x3 of an OHWI weight, once per tensor and version (aivc_conv_params.w_bf16x3).  The codec's
    side streams launch convolutions too, so the (one-off) split is closed with a device-wide wait like every other
    kernel-ready parameter (layers/_cache.py)."""
    import weakref
    key = id(w_ohwi)
    stamp = (w_ohwi.data_ptr(), w_ohwi._version)
    hit = _SPLIT_WEIGHTS.get(key)
    if hit is not None and hit[0]() is w_ohwi and hit[1] == stamp:
        return hit[2]
    co = w_ohwi.shape[0]
    k_total = w_ohwi.numel() // co
    out = torch.empty(co * k_total * 6, dtype=torch.uint8, device=w_ohwi.device)
    torch.cuda.synchronize(w_ohwi.device)
    call('aivc_split_weights_bf16x3', _p(w_ohwi), co, k_total, _p(out), _stream())
    torch.cuda.synchronize(w_ohwi.device)
    _SPLIT_WEIGHTS[key] = (weakref.ref(w_ohwi, lambda _r, k=key: _SPLIT_WEIGHTS.pop(k, None)), stamp, out)
    return out


_WINO_WEIGHTS = {}  # id(weight tensor) -> (weak reference, (data_ptr, version), U)


def winograd_weights(w_ohwi, transposed=False):
    """aivc_winograd_weights of an OHWI 3x3 weight ([co, 3, 3, ci] -> co * 16 * ci floats), aivc_winograd_weights_poly5 of a 5x5
    stride-2 one (4 ci virtual input channels) or, transposed, aivc_winograd_weights_tconv5 (4 co virtual output channels), once per
    tensor and version (aivc_conv_params.w_wino); closed with a device-wide wait like every other kernel-ready parameter."""
    import weakref
    key = (id(w_ohwi), bool(transposed))
    stamp = (w_ohwi.data_ptr(), w_ohwi._version)
    hit = _WINO_WEIGHTS.get(key)
    if hit is not None and hit[0]() is w_ohwi and hit[1] == stamp:
        return hit[2]
    co, k, _, ci = w_ohwi.shape
    fn = 'aivc_winograd_weights_tconv5' if transposed else ('aivc_winograd_weights_poly5' if k == 5 else 'aivc_winograd_weights')
    out = torch.empty(co * 16 * ci * (4 if k == 5 else 1), dtype=torch.float32, device=w_ohwi.device)  # (the kernels' staging order, AIVC_WINO_U_INDEX)
    torch.cuda.synchronize(w_ohwi.device)
    call(fn, _p(w_ohwi), co, ci, _p(out), _stream())
    torch.cuda.synchronize(w_ohwi.device)
    _WINO_WEIGHTS[key] = (weakref.ref(w_ohwi, lambda _r, k_=key: _WINO_WEIGHTS.pop(k_, None)), stamp, out)
    return out


def _winograd_covers(mode, k, stride, pad, c, co, act1, act2, h, w, tail=False):
    """include/aivc_hip.h: aivc_winograd_covers"""
    if tail or act1 == 3 or act2 == 3:
        return False
    if mode == abi.MODE_TCONV:  # transposed 5x5 stride 2, class by class: the size rule counts INPUT pixels
        return k == 5 and stride == 2 and c % 32 == 0 and co % 64 == 0 and (h * w >= abi.WINO_MIN_PIXELS_TCONV or WINO_ANY_SIZE)
    if mode != abi.MODE_CONV or co % 128:
        return False
    if k == 3 and stride == 1 and pad == 1 and c % 32 == 0:
        return h * w >= abi.WINO_MIN_PIXELS or WINO_ANY_SIZE
    if k == 5 and stride == 2 and pad == 2 and c >= 32 and c & (c - 1) == 0:  # polyphase form: the size rule counts OUTPUT pixels
        ho, wo = abi.conv_out_size(mode, h, w, k, stride, pad)
        return ho * wo >= abi.WINO_MIN_PIXELS or WINO_ANY_SIZE
    return False


WINO_ANY_SIZE = False  # tests: fp32w on images below AIVC_WINO_MIN_PIXELS too (aivc_conv_params.flags, AIVC_CONV_WINO_ANY_SIZE)
PRESPLIT_WEIGHTS = True  # bf16x3 mode: hand the kernels the split weights (False: they split in their K loop; same bits)


def conv2d(x, w_ohwi, bias=None, mode=abi.MODE_CONV, stride=1, pad=0, act1=0, act2=0, mul=None,
           res=None, algo=abi.ALGO_AUTO, gdn=None, tail=None):
    """x [n,h,w,c] -> y [n,ho,wo,co]; semantics of aivc_conv2d (include/aivc_hip.h).
    gdn = (beta_eff, gamma_eff, inverse) fuses the (inverse) GDN into the conv epilogue when the
    kernels can (all channels of a pixel in one tile), else it is issued as a second launch --
    bit-identical either way.
    tail = (w3 [co2,1,1,co], b3): a 1x1 conv applied to act1(conv + bias) in the same launch when the kernels
    can, else as a second launch (bit-identical); res / act2 then belong to the tail and y is [n,ho,wo,co2]."""
    if not isinstance(x, ImageStack) and getattr(x, '_aivc_cmap', None) == (0, 1, 2) and x.dim() == 4 and x.shape[-1] == 4 \
            and mode == abi.MODE_CONV and tail is None:
        st = ImageStack([x], x.shape[1], x.shape[2], x.device)  # a single float image (the prediction fed to g_a_ref)
        st._packed = x
        x = st
    if isinstance(x, ImageStack):
        y = None
        if tail is None and mode == abi.MODE_CONV:
            y = _conv_images(x, w_ohwi, bias, stride, pad, act1, act2, mul, res, algo, gdn)
        if y is not None:
            return y
        x = x.tensor()
    if tail is not None:
        return _conv2d_tail(x, w_ohwi, bias, stride, pad, act1, act2, res, algo, tail)
    x = _dev(x, torch.float32, 'x')
    n, h, w_, c = x.shape
    cmap = getattr(x, '_aivc_cmap', None)
    c_real = c if cmap is None else len(cmap)  # algorithmic FLOPs count real channels, not zero padding
    if c % 4:
        x = pad_channels(x, (c + 3) // 4 * 4)
        c = x.shape[-1]
    w_ohwi = _dev(w_ohwi, torch.float32, 'weight')
    co, k, _, cw = w_ohwi.shape
    if cw != c:
        raise AivcNativeError('conv2d: weight packed for %d stored channels, input has %d' % (cw, c))
    ho, wo = abi.conv_out_size(mode, h, w_, k, stride, pad)
    y = torch.empty((n, ho, wo, co), dtype=torch.float32, device=x.device)
    bias = _dev(bias, torch.float32, 'bias')
    mul = _dev(mul, torch.float32, 'mul')
    res = _dev(res, torch.float32, 'res')
    for t, nm in ((mul, 'mul'), (res, 'res')):
        if t is not None and tuple(t.shape) != tuple(y.shape):
            raise AivcNativeError('conv2d: %s shape %s != output shape %s' % (nm, tuple(t.shape), tuple(y.shape)))
    # 3-channel images stored as 4: the zero pad channels are exact no-ops the MFMA kernels may skip
    flags = abi.CONV_SPARSE4 if cmap is not None and all(ci % 4 != 3 for ci in cmap) else 0
    if WINO_ANY_SIZE:
        flags |= abi.CONV_WINO_ANY_SIZE
    if gdn is not None:
        g_beta, g_gamma, g_inv = gdn
        p = abi.ConvParams(mode, k, stride, pad, n, h, w_, c, ho, wo, co, act1, act2, algo, 2 if g_inv else 1, flags,
                           _p(x), _p(w_ohwi), _p(bias), _p(mul), _p(res), _p(y), _p(g_beta), _p(g_gamma))
        p.precision = PRECISION
        from ._lib import load
        if load()['aivc_conv2d_variant'](C.byref(p)) < 0:  # not fusable for this shape: two launches
            t = conv2d(x, w_ohwi, bias, mode=mode, stride=stride, pad=pad, algo=algo)
            return globals()['gdn'](t, g_beta, g_gamma, inverse=g_inv, res=res, algo=algo) if act1 == 0 and \
                act2 == 0 and mul is None else _unsupported_gdn_epilogue()
    else:
        p = abi.ConvParams(mode, k, stride, pad, n, h, w_, c, ho, wo, co, act1, act2, algo, 0, flags,
                           _p(x), _p(w_ohwi), _p(bias), _p(mul), _p(res), _p(y), None, None)
    p.precision = PRECISION
    if PRECISION == abi.PREC_BF16X3 and PRESPLIT_WEIGHTS and w_ohwi.is_contiguous() and (w_ohwi.numel() // co) % 32 == 0:
        from ._lib import load
        if load()['aivc_conv2d_variant'](C.byref(p)) >= 1000:  # a launch the mode covers
            p.w_bf16x3 = _p(split_weights_bf16x3(w_ohwi))
    if PRECISION == abi.PREC_FP32_WINO and _winograd_covers(mode, k, stride, pad, c, co, act1, act2, h, w_):
        p.w_wino = _p(winograd_weights(w_ohwi if w_ohwi.is_contiguous() else w_ohwi.contiguous(), transposed=mode == abi.MODE_TCONV))
    if PROFILE is None:
        call('aivc_conv2d', C.byref(p), _stream())
        return y
    # bench.py instrumentation: HIP events on the launch stream around this one kernel
    from ._lib import load
    variant = load()['aivc_conv2d_variant'](C.byref(p))
    taps = k * k
    pix = n * h * w_ if mode == abi.MODE_TCONV else n * ho * wo
    flops = 2.0 * taps * c_real * co * pix + (2.0 * co * co * n * ho * wo if gdn is not None else 0.0)
    if variant == 303:  # transposed 5x5 stride 2 by classes: 49 multiplications per 2 x 2 grid pixels (4 x 4 outputs) and channel pair instead of 100
        PROFILE_DIRECT_EQUIVALENT[0] += flops
        flops = 2.0 * 49 * c_real * co * n * ((h + 1) // 2) * ((w_ + 1) // 2)
        PROFILE_DIRECT_EQUIVALENT[1] += flops
    if variant == 302:  # 5x5 stride 2 in polyphase form: 49 multiplications per 2 x 2 outputs and channel pair instead of 100
        PROFILE_DIRECT_EQUIVALENT[0] += flops
        flops = 2.0 * 49 * c_real * co * n * ((ho + 1) // 2) * ((wo + 1) // 2)
        PROFILE_DIRECT_EQUIVALENT[1] += flops
    if variant == 301:
        # version 2 of the contract: what the matrix pipe EXECUTES (16 multiplications per tile of 2 x 2 outputs and channel
        # pair instead of 36) -- a roofline fraction is priced on issued work; the tap chain's count is kept beside it
        PROFILE_DIRECT_EQUIVALENT[0] += flops
        flops = 2.0 * 16 * c_real * co * n * ((h + 1) // 2) * ((w_ + 1) // 2)
        PROFILE_DIRECT_EQUIVALENT[1] += flops
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call('aivc_conv2d', C.byref(p), _stream())
    e1.record()
    PROFILE.append((variant, flops, e0, e1, (mode, k, stride, c_real, co, n, h, w_, gdn is not None)))
    return y


def _conv2d_tail(x, w_ohwi, bias, stride, pad, act1, act2, res, algo, tail):
    from ._lib import load
    w3, b3 = tail
    x = _dev(x, torch.float32, 'x')
    w_ohwi = _dev(w_ohwi, torch.float32, 'weight')
    w3 = _dev(w3, torch.float32, 'tail weight')
    n, h, w_, c = x.shape
    co, k, _, cw = w_ohwi.shape
    co2 = w3.shape[0]
    fused = c % 4 == 0 and cw == c and tuple(w3.shape) == (co2, 1, 1, co) and bias is not None and b3 is not None
    if fused:
        ho, wo = abi.conv_out_size(abi.MODE_CONV, h, w_, k, stride, pad)
        y = torch.empty((n, ho, wo, co2), dtype=torch.float32, device=x.device)
        bias, b3, res = _dev(bias, torch.float32, 'bias'), _dev(b3, torch.float32, 'tail bias'), _dev(res, torch.float32, 'res')
        if res is not None and tuple(res.shape) != tuple(y.shape):
            raise AivcNativeError('conv2d: res shape %s != output shape %s' % (tuple(res.shape), tuple(y.shape)))
        p = abi.ConvParams(abi.MODE_CONV, k, stride, pad, n, h, w_, c, ho, wo, co, act1, act2, algo, 0, 0,
                           _p(x), _p(w_ohwi), _p(bias), None, _p(res), _p(y), None, None, _p(w3), _p(b3), co2, 0)
        p.precision = PRECISION
        variant = load()['aivc_conv2d_variant'](C.byref(p))
        fused = variant >= 0
        if variant >= 1000 and PRESPLIT_WEIGHTS and w_ohwi.is_contiguous():
            p.w_bf16x3 = _p(split_weights_bf16x3(w_ohwi))
    if not fused:  # two launches, same arithmetic
        t = conv2d(x, w_ohwi, bias, stride=stride, pad=pad, act1=act1, algo=algo)
        return conv2d(t, w3, b3, res=res, act2=act2, algo=algo)
    if PROFILE is None:
        call('aivc_conv2d', C.byref(p), _stream())
        return y
    flops = 2.0 * (k * k * c * co + co * co2) * n * ho * wo
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call('aivc_conv2d', C.byref(p), _stream())
    e1.record()
    PROFILE.append((variant, flops, e0, e1, (abi.MODE_CONV, k, stride, c, co, n, h, w_, False)))
    return y


def _unsupported_gdn_epilogue():
    raise AivcNativeError('conv2d: fused gdn with act/mul epilogue needs a fusable shape')


def gdn_reparam(beta, gamma, beta_bound, gamma_bound, pedestal):
    beta = _dev(beta.detach(), torch.float32, 'beta')
    gamma = _dev(gamma.detach(), torch.float32, 'gamma')
    c = beta.shape[0]
    be = torch.empty_like(beta)
    ge = torch.empty_like(gamma)
    call('aivc_gdn_reparam', _p(beta), _p(gamma), c, float(beta_bound), float(gamma_bound),
         float(pedestal), _p(be), _p(ge), _stream())
    return be, ge


def selfcheck_gdn_math(n_div_pairs=1 << 34, seed=1, device=None):
    """-> (sqrt mismatches over every float of the safe range, division mismatches over n_div_pairs random pairs):
    the lean sequences of the fused GDN epilogues against the compiler's IEEE sqrt / division (aivc_selfcheck_gdn_math)"""
    out = torch.zeros(2, dtype=torch.int64, device=device or torch.device('cuda'))
    call('aivc_selfcheck_gdn_math', int(n_div_pairs), int(seed), _p(out), _stream())
    return tuple(int(v) for v in out.cpu())


def gdn(x, beta_eff, gamma_eff, inverse=False, res=None, algo=abi.ALGO_AUTO):
    c = x.shape[-1]
    mode = abi.MODE_IGDN if inverse else abi.MODE_GDN
    if c % 4 == 0:
        return conv2d(x, gamma_eff.reshape(c, 1, 1, c), beta_eff, mode=mode, res=res, algo=algo)
    # channel counts that are not a multiple of 4 (never the case in a real model): run on a zero
    # padded copy (extra channels: x = 0, gamma = 0, beta = 1) and drop the padding
    c4 = (c + 3) // 4 * 4
    g = torch.zeros((c4, c4), dtype=torch.float32, device=x.device)
    g[:c, :c] = gamma_eff
    b = torch.ones(c4, dtype=torch.float32, device=x.device)
    b[:c] = beta_eff
    y = conv2d(pad_channels(x, c4), g.reshape(c4, 1, 1, c4), b, mode=mode,
               res=None if res is None else pad_channels(res, c4), algo=algo)
    return y[..., :c].contiguous()


def yuv420_to_444(y, u, v, c_store=4, c_off=0, out=None):
    """planes [n,h,w] / [n,ceil(h/2),ceil(w/2)] (fp32 levels or uint8) -> NHWC [n,h,w,c_store]"""
    u8 = y.dtype == torch.uint8
    dt = torch.uint8 if u8 else torch.float32
    y, u, v = _dev(y, dt, 'y'), _dev(u, dt, 'u'), _dev(v, dt, 'v')
    n, h, w = y.shape
    if out is None:
        # the kernel writes channels c_off .. c_off + 2 (+ the zero pad channel): nothing else to clear when that is all
        full = c_off == 0 and c_store <= 4
        out = (torch.empty if full else torch.zeros)((n, h, w, c_store), dtype=torch.float32, device=y.device)
    zero_pad = 1 if out.shape[-1] >= c_off + 4 else 0
    hc, wc = (h + 1) // 2, (w + 1) // 2
    _hbm_profiled('yuv420_to_444', n * ((h * w + 2 * hc * wc) * (1 if u8 else 4) + h * w * 4 * (3 + zero_pad)),
                  lambda: call('aivc_yuv420u8_to_444' if u8 else 'aivc_yuv420_to_444', _p(y), _p(u), _p(v), n, h, w,
                               _p(out), out.shape[-1], c_off, zero_pad, _stream()))
    return out


_CONV_IMAGES_MAX = int(os.environ.get('AIVC_CONV_IMAGES_MAX', '2'))  # tuning aid: 0 disables aivc_conv_images
_IMAGES_UNFUSED_GDN = bool(os.environ.get('AIVC_IMAGES_UNFUSED_GDN'))  # tuning aid: the first layer's GDN as a launch of its own


class ImageStack:
    """The concatenation of up to 3 images (each stored as 3 real channels + a zero) as the first analysis conv sees
    it -- NOT materialised: conv2d() runs aivc_conv_images straight on the sources when the kernel covers the layer,
    and falls back to tensor() (aivc_pack_images) otherwise.  Quacks like the packed NHWC tensor for the few
    attributes the layers read (shape, device, _aivc_cmap)."""

    def __init__(self, parts, h, w, device):
        self.parts, self.h, self.w, self.device = list(parts), h, w, device
        self.n = next(p['y'].shape[0] if isinstance(p, dict) else p.shape[0] for p in parts if p is not None)
        self.shape = (self.n, h, w, 4 * len(self.parts))
        self.dtype = torch.float32
        self._aivc_cmap = tuple(4 * i + c for i in range(len(self.parts)) for c in range(3))
        self._packed = None

    def sources(self):
        """(ImageSrc array, tensors to keep alive)"""
        arr = (abi.ImageSrc * abi.MAX_IMAGES)()
        keep = []
        for i, p in enumerate(self.parts):
            if isinstance(p, dict):
                y, u, v = (_dev(p[k], torch.uint8, k) for k in 'yuv')
                keep += [y, u, v]
                arr[i].y, arr[i].u, arr[i].v = y.data_ptr(), u.data_ptr(), v.data_ptr()
            elif p is not None:
                f = _dev(p, torch.float32, 'image')
                keep.append(f)
                arr[i].f, arr[i].f_channels = f.data_ptr(), f.shape[-1]
        return arr, keep

    def tensor(self):
        if self._packed is None:
            self._packed = pack_images(self.parts, self.h, self.w, self.device)
        return self._packed


def _conv_images(x, w_ohwi, bias, stride, pad, act1, act2, mul, res, algo, gdn):
    """aivc_conv_images on an ImageStack; None when the library declines the layer (caller packs and convolves)"""
    from ._lib import load
    if mul is not None or res is not None or act2 or algo != abi.ALGO_AUTO:
        return None
    # three images (K = 300, 113 KB of LDS: one workgroup per CU) measured slower than pack + generic conv
    # (5.2 vs ~4.3 ms at 16 x 1080p), one and two images at par with the conv alone and without the packed tensor
    if len(x.parts) > _CONV_IMAGES_MAX:
        return None
    if gdn is not None and _IMAGES_UNFUSED_GDN:  # tuning aid: first layer without its fused GDN + a GDN-mode launch (bit identical)
        t = _conv_images(x, w_ohwi, bias, stride, pad, act1, act2, mul, res, algo, None)
        return None if t is None else globals()['gdn'](t, gdn[0], gdn[1], inverse=gdn[2])
    w_ohwi = _dev(w_ohwi, torch.float32, 'weight')
    co, k, _, cw = w_ohwi.shape
    n, h, w_, c = x.shape
    if cw != c:
        return None
    ho, wo = abi.conv_out_size(abi.MODE_CONV, h, w_, k, stride, pad)
    y = torch.empty((n, ho, wo, co), dtype=torch.float32, device=x.device)
    bias = _dev(bias, torch.float32, 'bias')
    g_beta = g_gamma = None
    gflag = 0
    if gdn is not None:
        g_beta, g_gamma, gflag = gdn[0], gdn[1], (2 if gdn[2] else 1)
    p = abi.ConvParams(abi.MODE_CONV, k, stride, pad, n, h, w_, c, ho, wo, co, act1, 0, algo, gflag, abi.CONV_SPARSE4,
                       None, _p(w_ohwi), _p(bias), None, None, _p(y), _p(g_beta), _p(g_gamma))
    arr, keep = x.sources()
    e0 = e1 = None
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = load()['aivc_conv_images'](arr, len(x.parts), C.byref(p), _stream())
    if rc == abi.ERR_UNSUPPORTED:
        return None
    if rc != 0:
        raise AivcNativeError('aivc_conv_images failed (%d)' % rc)
    if PROFILE is not None:
        e1.record()
        c_real = len(x._aivc_cmap)
        flops = 2.0 * k * k * c_real * co * n * ho * wo + (2.0 * co * co * n * ho * wo if gdn is not None else 0.0)
        PROFILE.append((191, flops, e0, e1, (abi.MODE_CONV, k, stride, c_real, co, n, h, w_, gdn is not None)))
    del keep
    return y


def pack_images(parts, h, w, device):
    """parts: up to 3 sources, each a dict of uint8 planes {'y','u','v'} ([n,h,w] / [n,ceil(h/2),ceil(w/2)]), a
    float NHWC tensor [n,h,w,c>=3] (its first 3 channels are taken) or None (zeros) -> NHWC [n,h,w,4*len(parts)]
    with every image stored as (c0, c1, c2, 0); carries the stored position of the real channels for the conv."""
    n = next(p['y'].shape[0] if isinstance(p, dict) else p.shape[0] for p in parts if p is not None)
    arr = (abi.ImageSrc * abi.MAX_IMAGES)()
    keep = []
    for i, p in enumerate(parts):
        if isinstance(p, dict):
            y, u, v = (_dev(p[k], torch.uint8, k) for k in 'yuv')
            keep += [y, u, v]
            arr[i].y, arr[i].u, arr[i].v = y.data_ptr(), u.data_ptr(), v.data_ptr()
        elif p is not None:
            f = _dev(p, torch.float32, 'image')
            keep.append(f)
            arr[i].f, arr[i].f_channels = f.data_ptr(), f.shape[-1]
    out = torch.empty((n, h, w, 4 * len(parts)), dtype=torch.float32, device=device)
    nb = n * h * w * (16 * len(parts) + sum(1.5 if isinstance(p, dict) else (12 if p is not None else 0) for p in parts))
    _hbm_profiled('pack_images', nb, lambda: call('aivc_pack_images', arr, len(parts), n, h, w, _p(out), _stream()))
    out._aivc_cmap = tuple(4 * i + c for i in range(len(parts)) for c in range(3))
    return out


def frame_to_yuv420(x, h, w, skip=None, want_float=True, want_u8=True):
    x = _dev(x, torch.float32, 'x')
    skip = _dev(skip, torch.float32, 'skip')
    n, hx, wx, cx = x.shape
    hc, wc = (h + 1) // 2, (w + 1) // 2
    dev = x.device
    f = [None] * 3
    b = [None] * 3
    if want_float:
        f = [torch.empty((n, h, w), dtype=torch.float32, device=dev),
             torch.empty((n, hc, wc), dtype=torch.float32, device=dev),
             torch.empty((n, hc, wc), dtype=torch.float32, device=dev)]
    if want_u8:
        b = [torch.empty((n, h, w), dtype=torch.uint8, device=dev),
             torch.empty((n, hc, wc), dtype=torch.uint8, device=dev),
             torch.empty((n, hc, wc), dtype=torch.uint8, device=dev)]
    # algorithmic bytes: the 3 real channels of x (and of skip) over the frame, the 4:2:0 planes out
    nb = n * (h * w * 3 * 4 * (2 if skip is not None else 1) + (h * w + 2 * hc * wc) * ((4 if want_float else 0) + (1 if want_u8 else 0)))
    _hbm_profiled('frame_to_yuv420', nb,
                  lambda: call('aivc_frame_to_yuv420', _p(x), n, hx, wx, cx, _p(skip), 0 if skip is None else skip.shape[-1],
                               h, w, _p(f[0]), _p(f[1]), _p(f[2]), _p(b[0]), _p(b[1]), _p(b[2]), _stream()))
    return tuple(f), tuple(b)


def downsample2x(x, ch0, nch):
    """NHWC x -> planar [n, nch, h//2, w//2] 2x2 means of channels ch0..ch0+nch-1 (OutputLayer)."""
    x = _dev(x, torch.float32, 'x')
    n, h, w, c = x.shape
    out = torch.empty((n, nch, h // 2, w // 2), dtype=torch.float32, device=x.device)
    call('aivc_downsample2x', _p(x), n, h, w, c, ch0, nch, _p(out), _stream())
    return out


def warp(x, flow):
    x, flow = _dev(x, torch.float32, 'x'), _dev(flow, torch.float32, 'flow')
    n, h, w, c = x.shape
    out = torch.empty_like(x)
    call('aivc_warp', _p(x), _p(flow), n, h, w, c, _p(out), _stream())
    return out


def warp_blend(mof, prev, nxt, h, w, frame_type, co=4, want_aux=False, rows=None):
    """rows = (row0, n_rows): only that band of the frame -- mof is then the band of the MOFNet output (its row 0 =
    frame row row0), prev / nxt stay whole frames, the outputs are [n, n_rows, w, co] (aivc_warp_blend_rows)."""
    mof, prev, nxt = (_dev(t, torch.float32, nm) for t, nm in ((mof, 'mof'), (prev, 'prev'), (nxt, 'next')))
    n, hm, wm, cm = mof.shape
    dev = mof.device
    row0, nr = (0, h) if rows is None else rows
    pred = torch.empty((n, nr, w, co), dtype=torch.float32, device=dev)
    skip = torch.empty_like(pred)
    xw = alpha = beta = None
    if want_aux:
        xw = torch.empty_like(pred)
        alpha = torch.empty((n, nr, w), dtype=torch.float32, device=dev)
        beta = torch.empty((n, nr, w), dtype=torch.float32, device=dev)
    # algorithmic bytes: the 6 MOFNet maps, 3 channels of each reference used, pred + skip out
    n_ref = 2 if int(frame_type) == FRAME_B else 1
    nb = n * nr * w * 4 * (6 + 3 * n_ref + 2 * co + ((co + 2) if want_aux else 0))
    _hbm_profiled('warp_blend', nb,
                  lambda: call('aivc_warp_blend_rows', _p(mof), hm, wm, cm, _p(prev), _p(nxt), prev.shape[-1], n, h, w,
                               int(row0), int(nr), int(frame_type), _p(pred), _p(skip), _p(xw), co, _p(alpha), _p(beta),
                               _stream()))
    return {'pred': pred, 'skip': skip, 'x_warp': xw, 'alpha': alpha, 'beta': beta}


def hyper_params(hs, c, h, w):
    hs = _dev(hs, torch.float32, 'hs')
    n, hh, wh, c2 = hs.shape
    if c2 != 2 * c:
        raise AivcNativeError('hyper_params: expected %d channels, got %d' % (2 * c, c2))
    mu = torch.empty((n, h, w, c), dtype=torch.float32, device=hs.device)
    sigma = torch.empty_like(mu)
    call('aivc_hyper_params', _p(hs), n, hh, wh, c, h, w, _p(mu), _p(sigma), _stream())
    return mu, sigma


def channel_gain(x, gain):
    x = _dev(x, torch.float32, 'x')
    gain = None if gain is None else _dev(gain.detach().reshape(-1), torch.float32, 'gain')
    out = torch.empty_like(x)
    call('aivc_channel_gain', _p(x), _p(gain), x.numel() // x.shape[-1], x.shape[-1], _p(out), _stream())
    return out


def gain_interp(g_r, g_t, lam):
    g_r = _dev(g_r.detach().reshape(-1), torch.float32, 'g_r')
    g_t = _dev(g_t.detach().reshape(-1), torch.float32, 'g_t')
    out = torch.empty_like(g_r)
    call('aivc_gain_interp', _p(g_r), _p(g_t), g_r.numel(), float(lam), _p(out), _stream())
    return out


def quantize_center(y, mu=None, gain_dec=None, want_yhat=True):
    y = _dev(y, torch.float32, 'y')
    mu = _dev(mu, torch.float32, 'mu')
    gain_dec = None if gain_dec is None else _dev(gain_dec.detach().reshape(-1), torch.float32, 'gain')
    q = torch.empty(y.shape, dtype=torch.int16, device=y.device)
    y_hat = torch.empty_like(y) if want_yhat else None
    call('aivc_quantize_center', _p(y), _p(mu), _p(gain_dec), y.numel() // y.shape[-1], y.shape[-1],
         _p(q), _p(y_hat), _stream())
    return q, y_hat


def dequantize(q, mu=None, gain_dec=None):
    q = _dev(q, torch.int16, 'q')
    mu = _dev(mu, torch.float32, 'mu')
    gain_dec = None if gain_dec is None else _dev(gain_dec.detach().reshape(-1), torch.float32, 'gain')
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    call('aivc_dequantize', _p(q), _p(mu), _p(gain_dec), q.numel() // q.shape[-1], q.shape[-1], _p(out),
         _stream())
    return out


# entropy model ---------------------------------------------------------------------------------
def balle_cdf_table(params, want_float=False):
    params = _dev(params, torch.float32, 'params')
    c = params.shape[0]
    table = torch.empty((c, abi.CDF_ROW), dtype=torch.int16, device=params.device)  # uint16 payload
    cdf = torch.empty((c, abi.LP), dtype=torch.float32, device=params.device) if want_float else None
    call('aivc_balle_cdf_table', _p(params), c, _p(table), _p(cdf), _stream())
    return (table, cdf) if want_float else table


def nonzero_flags(q):
    """q [n,h,w,c] int16 -> device uint8 [n, c] flags, one row per image (no host sync)"""
    q = _dev(q, torch.int16, 'q')
    n, c = q.shape[0], q.shape[-1]
    npix = q.numel() // (n * c)
    flags = torch.empty((n, c), dtype=torch.uint8, device=q.device)
    for i0 in range(0, n, 65535):  # one launch for the whole batch (grid.y = image)
        m = min(65535, n - i0)
        call('aivc_nonzero_maps_batch', q.data_ptr() + 2 * i0 * npix * c, m, npix, c, flags.data_ptr() + i0 * c, _stream())
    return flags


def laplace_cdf_rows(sigma, maps, out=None, row_off=0):
    """sigma [1,h,w,c] (one image) -> rows [(len(maps)*npix), CDF_ROW]; with `out` given the rows are
    written starting at row `row_off` of that (larger) tensor."""
    sigma = _dev(sigma, torch.float32, 'sigma')
    c = sigma.shape[-1]
    npix = sigma.numel() // c
    ml = abi.MapList.make(maps)
    if out is None:
        out = torch.empty((len(maps) * npix, abi.CDF_ROW), dtype=torch.int16, device=sigma.device)
        row_off = 0
    # fp64-VALU bound, not HBM bound: 514 expm1 evaluations per coded position; the profile counts CDF POINTS
    _hbm_profiled('cdf_points:laplace_cdf_rows', len(maps) * npix * abi.LP,
                  lambda: call('aivc_laplace_cdf_rows', _p(sigma), npix, c, C.byref(ml),
                               out.data_ptr() + 2 * row_off * abi.CDF_ROW, _stream()))
    return out


def laplace_cdf_windows(sigma, maps, out=None, row_off=0):
    """Like laplace_cdf_rows, restricted to the decoder's fast-path window: -> (win [n_pos, CDF_WIN] int16,
    sigma_pos [n_pos] float32); with `out` = (win, sigma_pos) given, written from position `row_off` on."""
    sigma = _dev(sigma, torch.float32, 'sigma')
    c = sigma.shape[-1]
    npix = sigma.numel() // c
    ml = abi.MapList.make(maps)
    if out is None:
        out = (torch.empty((len(maps) * npix, abi.CDF_WIN), dtype=torch.int16, device=sigma.device),
               torch.empty(len(maps) * npix, dtype=torch.float32, device=sigma.device))
        row_off = 0
    win, sp = out
    _hbm_profiled('cdf_points:laplace_cdf_windows', len(maps) * npix * abi.CDF_WIN,
                  lambda: call('aivc_laplace_cdf_windows', _p(sigma), npix, c, C.byref(ml),
                               win.data_ptr() + 2 * row_off * abi.CDF_WIN, sp.data_ptr() + 4 * row_off, _stream()))
    return out


def frame_maps_to_device(maps_list, npix, device):
    """device image of the aivc_frame_maps table of a frame batch (pinned staging, asynchronous copy)
    -> (uint8 CUDA tensor [n, 272], [pos_off per frame], total coded positions)"""
    tab, offs, total = abi.frame_maps_table(maps_list, npix)
    host = _pinned_staging(tab.size)
    host.numpy()[:] = tab.reshape(-1)
    dev = torch.empty(tab.size, dtype=torch.uint8, device=device)
    dev.copy_(host, non_blocking=True)
    _pinned_release(host)
    return dev, offs, total


def laplace_cdf_windows_batch(sigma, maps_list, out, table=None):
    """the windows + sigma of every coded position of a frame batch in ONE launch: sigma [n,h,w,c], maps_list[f] the coded
    channels of frame f, out = (win [>= total, CDF_WIN] int16, sigma_pos [>= total]) -> ([pos_off per frame], table)"""
    sigma = _dev(sigma, torch.float32, 'sigma')
    n, c = sigma.shape[0], sigma.shape[-1]
    npix = sigma.numel() // (n * c)
    table = table or frame_maps_to_device(maps_list, npix, sigma.device)
    dev, offs, total = table
    win, sp = out
    assert win.shape[0] >= total and sp.shape[0] >= total
    mx = max((len(m) for m in maps_list), default=0)
    if total:
        _hbm_profiled('cdf_points:laplace_cdf_windows', total * abi.CDF_WIN,
                      lambda: call('aivc_laplace_cdf_windows_batch', _p(sigma), n, npix, c, _p(dev), mx, _p(win), _p(sp), _stream()))
    return offs, table


def laplace_bounds_batch(sigma, q, maps_list):
    """-> (bounds int32 [total], [pos_off per frame]) of a frame batch in one launch"""
    sigma, q = _dev(sigma, torch.float32, 'sigma'), _dev(q, torch.int16, 'q')
    n, c = sigma.shape[0], sigma.shape[-1]
    npix = sigma.numel() // (n * c)
    dev, offs, total = frame_maps_to_device(maps_list, npix, sigma.device)
    bounds = torch.empty(max(total, 1), dtype=torch.int32, device=sigma.device)
    mx = max((len(m) for m in maps_list), default=0)
    if total:
        call('aivc_laplace_bounds_batch', _p(sigma), _p(q), n, npix, c, _p(dev), mx, _p(bounds), _stream())
    return bounds, offs


def table_bounds_batch(table, q):
    """q [n,h,w,c] -> bounds int32 [n, c * npix] (every channel of every frame, pmf mode) in one launch"""
    q = _dev(q, torch.int16, 'q')
    n, c = q.shape[0], q.shape[-1]
    npix = q.numel() // (n * c)
    bounds = torch.empty((n, c * npix), dtype=torch.int32, device=q.device)
    call('aivc_table_bounds_batch', _p(table), _p(q), n, npix, c, _p(bounds), _stream())
    return bounds


def scatter_symbols_batch(sym, maps_list, n, npix, c, table=None):
    """sym: the decoded symbols of the batch in stream order (frame f's at its pos_off) -> q int16 [n, npix, c]"""
    dev_ = sym.device
    table = table or frame_maps_to_device(maps_list, npix, dev_)
    q = torch.empty((n, npix, c), dtype=torch.int16, device=dev_)
    call('aivc_scatter_symbols_batch', _p(sym), n, npix, c, _p(table[0]), _p(q), _stream())
    return q


def laplace_bounds(sigma, q, maps):
    sigma, q = _dev(sigma, torch.float32, 'sigma'), _dev(q, torch.int16, 'q')
    c = sigma.shape[-1]
    npix = sigma.numel() // c
    ml = abi.MapList.make(maps)
    bounds = torch.empty(len(maps) * npix, dtype=torch.int32, device=sigma.device)
    call('aivc_laplace_bounds', _p(sigma), _p(q), npix, c, C.byref(ml), _p(bounds), _stream())
    return bounds


def table_bounds(table, q):
    q = _dev(q, torch.int16, 'q')
    c = q.shape[-1]
    npix = q.numel() // c
    bounds = torch.empty(c * npix, dtype=torch.int32, device=q.device)
    call('aivc_table_bounds', _p(table), _p(q), npix, c, _p(bounds), _stream())
    return bounds


# ---- rate estimation (logging only; csrc/rate.hip) ---------------------------------------------------------
def _rate_scratch(device, out=None):
    lanes = torch.empty(abi.RATE_LANES, dtype=torch.float64, device=device)
    return lanes, (torch.empty(1, dtype=torch.float64, device=device) if out is None else out)


def bounds_rate(bounds, out=None):
    """packed CDF bounds (laplace_bounds / table_bounds) -> device fp64 [1]: the bits the range coder pays for them
    (sum of -log2((c_hi - c_lo) / 2^16), fixed summation order).  No sync.  out: a 1-element fp64 view to fill."""
    bounds = _dev(bounds, torch.int32, 'bounds')
    lanes, out = _rate_scratch(bounds.device, out)
    call('aivc_bounds_rate', _p(bounds), bounds.numel(), _p(lanes), _p(out), _stream())
    return out


def rate_bits(prob, p_min, p_max):
    """EntropyCoder.forward (src/layers/entropy_coding/entropy_coder.py:25-30): -log2(clamp(prob)) -> (rate like prob,
    device fp64 [1] sum)"""
    prob = _dev(prob, torch.float32, 'prob')
    rate = torch.empty_like(prob)
    lanes, out = _rate_scratch(prob.device)
    call('aivc_rate_bits', _p(prob), prob.numel(), float(p_min), float(p_max), _p(rate), _p(lanes), _p(out), _stream())
    return rate, out


def laplace_prob(y, mu, sigma):
    """P(bin of y) under Laplace(mu, sigma / sqrt(2)) elementwise (mu None = 0), ParametricPdf.forward"""
    y = _dev(y, torch.float32, 'y')
    sigma = _dev(sigma, torch.float32, 'sigma')
    mu = _dev(mu, torch.float32, 'mu')
    if sigma.shape != y.shape or (mu is not None and mu.shape != y.shape):
        raise AivcNativeError('aivc_amd.ops.laplace_prob: y, mu and sigma must have one shape')
    prob = torch.empty_like(y)
    call('aivc_laplace_prob', _p(y), _p(mu), _p(sigma), y.numel(), _p(prob), _stream())
    return prob


def table_prob(x, cdf_f32):
    """x [b, c, h, w] integer-valued in [-256, 256], cdf_f32 [c, 514] (balle_cdf_table(want_float=True)) -> P(bin of x)"""
    x = _dev(x, torch.float32, 'x')
    cdf_f32 = _dev(cdf_f32, torch.float32, 'cdf_f32')
    b, c, h, w = x.shape
    if tuple(cdf_f32.shape) != (c, abi.LP):
        raise AivcNativeError('aivc_amd.ops.table_prob: cdf table %s for %d channels' % (tuple(cdf_f32.shape), c))
    prob = torch.empty_like(x)
    call('aivc_table_prob', _p(x), _p(cdf_f32), x.numel(), h * w, c, _p(prob), _stream())
    return prob


_STAGING = []  # [(event or None, pinned uint8 tensor)]


def _pinned_staging(n):
    """pinned host buffer of >= n bytes whose previous H2D copy (if any) has completed"""
    for i, (ev, buf) in enumerate(_STAGING):
        if buf.numel() >= n and (ev is None or ev.query()):
            _STAGING.pop(i)
            return buf[:n]
    return torch.empty(max(1 << 20, 1 << (int(n) - 1).bit_length()), dtype=torch.uint8, pin_memory=True)[:n]


def _pinned_release(view):
    ev = torch.cuda.Event()
    ev.record()
    base = view._base if view._base is not None else view
    _STAGING.append((ev, base))


def range_encode(bounds_list, streams=None):
    """bounds_list: int32 CUDA tensors, one independent stream each (any number; launched 64 at a
    time, one wavefront per stream).  Returns (out uint8 tensor, lens int32 tensor [n], [(offset,
    capacity)]) -- all on device, no sync.
    streams: optional HIP streams; the launches (a kernel lasts as long as its longest stream) then
    run concurrently on them, forked from / joined back into the current stream with events."""
    n = len(bounds_list)
    dev = bounds_list[0].device
    # views that already sit back to back in one tensor (the batched bounds kernels) need no concatenation
    base = bounds_list[0]._base if bounds_list[0]._base is not None else bounds_list[0]
    pos, packed = bounds_list[0].storage_offset(), True
    for b in bounds_list:
        packed = packed and (b._base if b._base is not None else b) is base and b.storage_offset() == pos and b.is_contiguous()
        pos += b.numel()
    if packed:
        allb = base.reshape(-1)[bounds_list[0].storage_offset():pos]
    else:
        allb = bounds_list[0] if n == 1 else torch.cat(bounds_list)
    in_offs, out_offs, in_off, out_off = [], [], 0, 0
    for b in bounds_list:
        cap = (16 + 3 * b.numel() + 3) // 4 * 4
        in_offs.append(in_off)
        out_offs.append((out_off, cap))
        in_off += b.numel()
        out_off += cap
    out = torch.empty(out_off, dtype=torch.uint8, device=dev)
    lens = torch.empty(n, dtype=torch.int32, device=dev)
    fork = streams is not None and len(streams) > 0 and n > abi.RC_MAX_STREAMS
    cur, ev0 = None, None
    if fork:
        cur = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        for t in (allb, out, lens):
            for st in streams:
                t.record_stream(st)
    for gi, start in enumerate(range(0, n, abi.RC_MAX_STREAMS)):
        batch = abi.RcBatch()
        cnt = 0
        for i in range(start, min(n, start + abi.RC_MAX_STREAMS)):
            s_ = batch.s[cnt]
            s_.in_off, s_.out_off, s_.n_sym, s_.out_cap = in_offs[i], out_offs[i][0], bounds_list[i].numel(), out_offs[i][1]
            cnt += 1
        batch.n_streams = cnt
        if fork:
            st = streams[gi % len(streams)]
            st.wait_event(ev0)
            call('aivc_range_encode', _p(allb), C.byref(batch), _p(out), lens.data_ptr() + 4 * start, st.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(st)
            cur.wait_event(ev)
        else:
            call('aivc_range_encode', _p(allb), C.byref(batch), _p(out), lens.data_ptr() + 4 * start, _stream())
    return out, lens, out_offs


def range_decode(payloads, rows, row_offs, n_syms, planes, sigma_pos=None, want_bits=False, flat=False):
    """Decode len(payloads) independent streams concurrently (one wavefront each, <= 64 per launch).
    rows: ONE int16 CUDA tensor [n_rows, CDF_ROW] holding every stream's CDF rows; stream i starts at
    row row_offs[i] and uses one row per symbol (planes[i] == 0) or row i // planes[i] (pmf tables).
    With sigma_pos (float32 [n_rows]) `rows` holds the 64-entry windows of laplace_cdf_windows instead
    ([n_rows, CDF_WIN]; planes must be 0).
    Returns a list of int16 CUDA tensors (uint16 payload: symbols 0..512); with want_bits also an int32 CUDA tensor [n]
    of the bits each stream consumed (include/aivc_hip.h: len(payload) == (bits + 2 + 7) // 8 for an intact stream)."""
    n = len(payloads)
    dev = rows.device
    offs, total = [], 0
    for pl in payloads:
        offs.append(total)
        total += len(pl) + (-len(pl)) % 4 + 8
    host = _pinned_staging(max(total, 4))
    hv = host.numpy()
    hv[:] = 0
    for pl, o in zip(payloads, offs):
        if len(pl):
            hv[o:o + len(pl)] = np.frombuffer(pl, np.uint8)
    dbytes = torch.empty(host.shape, dtype=torch.uint8, device=dev)
    dbytes.copy_(host, non_blocking=True)  # pinned source: truly asynchronous, the host never waits
    _pinned_release(host)
    sym_total = int(sum(n_syms))
    sym = torch.empty(max(sym_total, 1), dtype=torch.int16, device=dev)
    outs, so = [], 0
    bits = torch.empty(n, dtype=torch.int32, device=dev) if want_bits else None
    for start in range(0, n, abi.RC_MAX_STREAMS):
        bp = None if bits is None else bits.data_ptr() + 4 * start
        batch = abi.RcBatch()
        cnt = 0
        for i in range(start, min(n, start + abi.RC_MAX_STREAMS)):
            s_ = batch.s[cnt]
            s_.in_off, s_.out_off, s_.row_off = offs[i], so, row_offs[i]
            s_.n_sym, s_.in_len, s_.plane = n_syms[i], len(payloads[i]), planes[i]
            outs.append(sym[so:so + n_syms[i]])
            so += n_syms[i]
            cnt += 1
        batch.n_streams = cnt
        if sigma_pos is None:
            call('aivc_range_decode', _p(dbytes), _p(rows), C.byref(batch), _p(sym), bp, _stream())
        else:
            call('aivc_range_decode_windows', _p(dbytes), _p(rows), _p(sigma_pos), C.byref(batch), _p(sym), bp, _stream())
    if flat:  # the one tensor the streams' symbols sit in, back to back in stream order
        outs = sym
    return (outs, bits) if want_bits else outs


def scatter_symbols(sym, npix, c, maps):
    ml = abi.MapList.make(maps)
    dev = sym.device if sym is not None else torch.device('cuda')
    q = torch.empty((npix, c), dtype=torch.int16, device=dev)
    call('aivc_scatter_symbols', _p(sym), npix, c, C.byref(ml), _p(q), _stream())
    return q


# ---- quality metrics (fp64 planes [n, h, w]) ------------------------------------------------------
def _metrics_ws(n, h, w, dev):
    nbytes = load()['aivc_metrics_workspace'](int(n), int(h), int(w))
    return torch.empty(nbytes // 8, dtype=torch.float64, device=dev)


def ssim_means(a, b, win, c1, c2):
    """a, b: float64 CUDA planes [n,h,w]; win: 1-D normalised window (host sequence, len <= 11).
    -> float64 CUDA tensor [n, 2]: (mean SSIM, mean contrast-structure) of one scale."""
    a, b = _dev(a, torch.float64, 'a'), _dev(b, torch.float64, 'b')
    n, h, w = a.shape
    wn = np.ascontiguousarray(np.asarray(win, np.float64))
    out = torch.empty((n, 2), dtype=torch.float64, device=a.device)
    ws = _metrics_ws(n, h, w, a.device)
    call('aivc_ssim_means', _p(a), _p(b), n, h, w, wn.ctypes.data, len(wn), float(c1), float(c2), _p(ws), _p(out), _stream())
    return out


def pool2x2(x, edge):
    """float64 planes [n,h,w] -> [n, ceil(h/2), ceil(w/2)] 2x2 means; edge 0: mirror past the end without
    repeating the border (torch ReflectionPad2d), 1: repeat it (scipy 'reflect')."""
    x = _dev(x, torch.float64, 'x')
    n, h, w = x.shape
    out = torch.empty((n, (h + 1) // 2, (w + 1) // 2), dtype=torch.float64, device=x.device)
    call('aivc_pool2x2', _p(x), n, h, w, int(edge), _p(out), _stream())
    return out


def sq_err(a, b):
    """-> float64 CUDA tensor [1]: sum of squared differences of two float64 tensors of equal size"""
    a, b = _dev(a, torch.float64, 'a'), _dev(b, torch.float64, 'b')
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    ws = _metrics_ws(1, 16, 16, a.device)
    call('aivc_sq_err', _p(a), _p(b), a.numel(), _p(ws), _p(out), _stream())
    return out
