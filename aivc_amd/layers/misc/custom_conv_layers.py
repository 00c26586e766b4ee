"""Conv building blocks with the reference's names and state_dict layout
(src/layers/misc/custom_conv_layers.py): CustomConvLayer, UpscalingLayer, ChengResBlock, ResBlock.

Each block is a thin description (torch sub-modules hold the parameters exactly where the
reference's pickles put them); forward() lowers it to fused aivc_conv2d launches:
  CustomConvLayer  = replicate-pad conv + bias + {leaky, relu} fused, GDN/IGDN as a second launch
  UpscalingLayer   = transposed conv (4 output-parity sub-convolutions) + the same epilogues
  ChengResBlock    = 2-3 launches with the skip connection added in the last epilogue
  ResBlock         = 2 launches (relu fused; `relu(x + f(x))` in the second epilogue)
"""
import numpy as np
import torch
from torch import nn
from torch.nn import Conv2d, ConvTranspose2d, LeakyReLU, ReLU, ReplicationPad2d, Sequential

from ... import abi, bands, ops
from .._cache import cached
from .misc_layers import GDN


def _act_of(module):
    if module is None:
        return abi.ACT_NONE
    if isinstance(module, LeakyReLU):
        if abs(module.negative_slope - 0.01) > 1e-12:
            raise NotImplementedError('LeakyReLU slope %r' % module.negative_slope)
        return abi.ACT_LEAKY
    if isinstance(module, ReLU):
        return abi.ACT_RELU
    if isinstance(module, nn.Sigmoid):
        return abi.ACT_SIGMOID
    raise NotImplementedError('unsupported non-linearity %r' % type(module).__name__)


def _sq(v):
    return v[0] if isinstance(v, (tuple, list)) else v


def packed_conv(conv, c_store, device, cmap=None):
    """(OHWI weight, bias) of a Conv2d / ConvTranspose2d on `device`, input channels padded to c_store.
    cmap: stored position of every real input channel (default: the first ones) -- lets a caller feed
    images whose 3-channel planes are each padded to 4 ([y,u,v,0, y,u,v,0, ...]) without a repack."""
    transposed = isinstance(conv, ConvTranspose2d)

    def build():
        w = ops.pack_weight(conv.weight.to(device, torch.float32), None, transposed=transposed)
        ci = w.shape[3]
        wp = torch.zeros(w.shape[:3] + (c_store,), dtype=torch.float32, device=device)
        wp[..., list(cmap) if cmap is not None else slice(0, ci)] = w
        b = None if conv.bias is None else conv.bias.detach().to(device, torch.float32).contiguous()
        return wp.contiguous(), b
    params = (conv.weight,) if conv.bias is None else (conv.weight, conv.bias)
    return cached(conv, ('w', c_store, str(device), None if cmap is None else tuple(cmap)), params, build)


def run_conv(conv, x, pad, act1=abi.ACT_NONE, act2=abi.ACT_NONE, res=None, mul=None, gdn=None, tail=None):
    """x NHWC -> NHWC through one aivc_conv2d launch for a torch Conv2d / ConvTranspose2d.
    gdn: optional GDN module applied to the conv output (fused into the epilogue when possible).
    tail: optional 1x1 Conv2d applied to act1(conv(x)) in the same launch when possible; res / act2 then
    belong to the tail.
    x may be a row band of the map (aivc_amd/bands.py: one frame over the ranks of a unit group): the same launch then
    runs on this rank's slab (band + halo rows fetched from the neighbours) and a band comes back."""
    if isinstance(x, (bands.Band, bands.BandImages)):
        transposed = isinstance(conv, ConvTranspose2d)

        def launch(xs, rs, ms):
            return run_conv(conv, xs, pad, act1=act1, act2=act2, res=rs, mul=ms, gdn=gdn, tail=tail)
        return x.ctx.conv(launch, x, abi.MODE_TCONV if transposed else abi.MODE_CONV, _sq(conv.kernel_size),
                          _sq(conv.stride), 0 if transposed else pad,
                          (tail if tail is not None else conv).out_channels, res=res, mul=mul)
    c_store = (x.shape[-1] + 3) // 4 * 4
    w, b = packed_conv(conv, c_store, x.device, getattr(x, '_aivc_cmap', None))
    if tail is not None:
        if gdn is not None or mul is not None or isinstance(conv, ConvTranspose2d) or _sq(tail.kernel_size) != 1:
            raise NotImplementedError('fused tail: a plain Conv2d followed by a 1x1 Conv2d')
        w3, b3 = packed_conv(tail, (conv.out_channels + 3) // 4 * 4, x.device, None)
        return ops.conv2d(x, w, b, stride=_sq(conv.stride), pad=pad, act1=act1, act2=act2, res=res, tail=(w3, b3))
    g = None
    if gdn is not None:
        if gdn.beta.shape[0] % 4:  # exotic channel count: padded stand-alone path
            return gdn.forward_nhwc(run_conv(conv, x, pad), res=res)
        be, ge = gdn.effective_params(x.device)
        g = (be, ge, bool(gdn.inverse))
    if isinstance(conv, ConvTranspose2d):
        k = _sq(conv.kernel_size)
        if _sq(conv.stride) != 2 or _sq(conv.output_padding) != 1 or _sq(conv.padding) != (k + 1) // 2 - 1:
            raise NotImplementedError('only the reference UpscalingLayer geometry is implemented')
        return ops.conv2d(x, w, b, mode=abi.MODE_TCONV, stride=2, act1=act1, act2=act2, res=res, mul=mul, gdn=g)
    if _sq(conv.padding) != 0 or _sq(conv.dilation) != 1 or conv.groups != 1:
        raise NotImplementedError('Conv2d with built-in padding/dilation/groups is not used by the codec')
    return ops.conv2d(x, w, b, stride=_sq(conv.stride), pad=pad, act1=act1, act2=act2, res=res, mul=mul, gdn=g)


def _split_nl(seq):
    """(non-linearity module or None) registered as `non_linearity` in a reference Sequential."""
    return seq._modules.get('non_linearity', None)


class CustomConvLayer(nn.Module):
    """ReplicationPad2d(k//2) + Conv2d(k, stride) + non-linearity in {gdn, gdn_inverse, leaky_relu,
    relu, no}.  Reference: src/layers/misc/custom_conv_layers.py:129-180."""

    def __init__(self, k_size=5, in_ft=64, out_ft=64, flag_bias=True, non_linearity='leaky_relu',
                 conv_stride=1, padding_mode='replicate'):
        super().__init__()
        if padding_mode != 'replicate':
            raise NotImplementedError(padding_mode)
        self.layers = Sequential(ReplicationPad2d(int(np.floor(k_size / 2))),
                                 Conv2d(in_ft, out_ft, k_size, stride=conv_stride, bias=flag_bias))
        _add_non_linearity(self.layers, non_linearity, out_ft)

    def forward_nhwc(self, x, res=None):
        pad_mod, conv = self.layers[0], self.layers[1]
        pad = _sq(pad_mod.padding)
        nl = _split_nl(self.layers)
        if isinstance(nl, GDN):
            return run_conv(conv, x, pad, res=res, gdn=nl)
        return run_conv(conv, x, pad, act1=_act_of(nl), res=res)

    def forward(self, x):
        return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(x)))


class UpscalingLayer(nn.Module):
    """x2 upsampling: ConvTranspose2d(k, stride 2, padding int((1+k)/2-1), output_padding 1) +
    non-linearity.  Reference: src/layers/misc/custom_conv_layers.py:183-253."""

    def __init__(self, k_size=5, in_ft=64, out_ft=64, flag_bias=True, non_linearity='leaky_relu',
                 mode='transposed', flag_first_layer=False):
        super().__init__()
        if mode == 'transposed_no_bias':
            flag_bias = False
        self.layers = Sequential(ConvTranspose2d(in_ft, out_ft, k_size, stride=2,
                                                 padding=int(((1 + k_size) / 2) - 1), output_padding=1,
                                                 bias=flag_bias))
        _add_non_linearity(self.layers, non_linearity, out_ft)

    def forward_nhwc(self, x, res=None):
        nl = _split_nl(self.layers)
        if isinstance(nl, GDN):
            return run_conv(self.layers[0], x, 0, res=res, gdn=nl)
        return run_conv(self.layers[0], x, 0, act1=_act_of(nl), res=res)

    def forward(self, x):
        return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(x)))


def _add_non_linearity(seq, name, ch):
    if name == 'gdn':
        seq.add_module('non_linearity', GDN(ch, inverse=False))
    elif name == 'gdn_inverse':
        seq.add_module('non_linearity', GDN(ch, inverse=True))
    elif name == 'leaky_relu':
        seq.add_module('non_linearity', LeakyReLU())
    elif name == 'relu':
        seq.add_module('non_linearity', ReLU())


class ChengResBlock(nn.Module):
    """Residual blocks of Cheng et al. 2019 as arranged by the reference
    (src/layers/misc/custom_conv_layers.py:21-109): `plain` x + f(x); `down` 1x1-s2(x) + f(x);
    `up_tconv` tconv3(x) + f(x)."""

    def __init__(self, nb_ft, mode='plain'):
        super().__init__()
        self.mode = mode
        if mode == 'plain':
            self.layers = Sequential(CustomConvLayer(3, nb_ft, nb_ft, non_linearity='leaky_relu'),
                                     CustomConvLayer(3, nb_ft, nb_ft, non_linearity='leaky_relu'))
        elif mode == 'down':
            self.layers = Sequential(CustomConvLayer(3, nb_ft, nb_ft, non_linearity='leaky_relu', conv_stride=2),
                                     CustomConvLayer(3, nb_ft, nb_ft, non_linearity='gdn'))
            self.aux_layer = Conv2d(nb_ft, nb_ft, 1, stride=2)
        elif mode == 'up_tconv':
            self.layers = Sequential(UpscalingLayer(3, nb_ft, nb_ft, non_linearity='leaky_relu'),
                                     CustomConvLayer(3, nb_ft, nb_ft, non_linearity='gdn_inverse'))
            self.aux_layer = UpscalingLayer(3, nb_ft, nb_ft, non_linearity='no')
        else:
            raise ValueError(mode)

    def forward_nhwc(self, x):
        if self.mode == 'plain':
            skip = x
        elif isinstance(self.aux_layer, Conv2d):
            skip = run_conv(self.aux_layer, x, 0)
        else:
            skip = self.aux_layer.forward_nhwc(x)
        h = self.layers[0].forward_nhwc(x)
        return self.layers[1].forward_nhwc(h, res=skip)

    def forward(self, x):
        return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(x)))


class ResBlock(nn.Module):
    """relu(x + conv(relu(conv(x)))) with replicate padding (src/layers/misc/custom_conv_layers.py:112-126)."""

    def __init__(self, k_size, nb_ft):
        super().__init__()
        p = int(np.floor(k_size / 2))
        self.layers = Sequential(ReplicationPad2d(p), Conv2d(nb_ft, nb_ft, k_size), ReLU(),
                                 ReplicationPad2d(p), Conv2d(nb_ft, nb_ft, k_size))

    def forward_nhwc(self, x):
        pad = _sq(self.layers[0].padding)
        h = run_conv(self.layers[1], x, pad, act1=abi.ACT_RELU)
        return run_conv(self.layers[4], h, pad, res=x, act2=abi.ACT_RELU)

    def forward(self, x):
        return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(x)))
