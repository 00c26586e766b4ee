"""torch-CPU conv backend of the CPU ORACLE -- TEST / BENCHMARK INFRASTRUCTURE ONLY (bench.py's `cpu_baseline`).

The reference runs its transforms as torch modules on the host when launched with --cpu
(src/encode.py:85-93, src/model_mngt/model_management.py): ReplicationPad2d + Conv2d
(src/layers/misc/custom_conv_layers.py:129-180), ConvTranspose2d (k, stride 2, padding int((1 + k) / 2 - 1),
output_padding 1, :183-253), GDN as a 1x1 conv on x^2 (src/layers/misc/misc_layers.py:131-149), the residual /
attention compositions (custom_conv_layers.py:21-126, attention.py:22-97).  This module evaluates the SAME exported
layer specs (oracle/spec.py) with those torch ops on CPU tensors, so the baseline times what the reference's --cpu
path spends its time in -- ATen / oneDNN convolutions on all host cores -- instead of the parity checker's scalar
fmaf chains (oracle/aivc_oracle.c, ~1 % of the host's fp32 peak).  Everything around the transforms (4:2:0 <-> 4:4:4,
warp, hyperprior split, CDF build, range coder, container) stays on the oracle's C: the reference does those on
the CPU as well, with far more work per symbol (a [C, H, W, 514] fp32 CDF tensor per latent).

Not a parity path: torch's summation order differs from the arithmetic contract of include/aivc_hip.h, results
agree with the oracle to ~1e-5 relative.  Encode and decode through THIS backend are consistent with each other
(same ops, same thread count), which `closed_loop` in the bench line records."""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O

_W = {}  # id(array) -> torch tensor (weights converted once; the arrays live in the spec for the whole run)


def _t(a):
    if a is None:
        return None
    key = id(a)
    hit = _W.get(key)
    if hit is None or hit[0] is not a:
        hit = _W[key] = (a, torch.from_numpy(np.ascontiguousarray(a, np.float32)))
    return hit[1]


def _act(nl, x):
    if nl in (None, 'no'):
        return x
    if nl == 'leaky_relu':
        return F.leaky_relu(x, 0.01)
    if nl == 'relu':
        return F.relu(x)
    if nl == 'sigmoid':
        return torch.sigmoid(x)
    raise ValueError(nl)


def _gdn(g, x):
    """src/layers/misc/misc_layers.py:113-154"""
    key = ('gdn', id(g['beta']))
    hit = _W.get(key)
    if hit is None or hit[0] is not g['beta']:
        beta = torch.from_numpy(g['beta'])
        gamma = torch.from_numpy(g['gamma'])
        be = torch.clamp_min(beta, g['beta_bound']) ** 2 - g['pedestal']
        ge = torch.clamp_min(gamma, g['gamma_bound']) ** 2 - g['pedestal']
        hit = _W[key] = (g['beta'], be.contiguous(), ge.reshape(ge.shape[0], ge.shape[1], 1, 1).contiguous())
    norm = torch.sqrt(F.conv2d(x * x, hit[2], hit[1]))
    return x * norm if g['inverse'] else x / norm


def _conv(x, w, b, k, stride):
    if k > 1:
        x = F.pad(x, (k // 2,) * 4, mode='replicate')
    return F.conv2d(x, _t(w), _t(b), stride=stride)


def _run(spec, x):
    t = spec['type']
    if t == 'Sequential':
        for s in spec['layers']:
            x = _run(s, x)
        return x
    if t == 'CustomConvLayer':
        y = _conv(x, spec['weight'], spec.get('bias'), spec['k'], spec['stride'])
        return _gdn(spec['gdn'], y) if spec['nl'] in ('gdn', 'gdn_inverse') else _act(spec['nl'], y)
    if t == 'UpscalingLayer':
        k = spec['k']
        y = F.conv_transpose2d(x, _t(spec['weight']), _t(spec.get('bias')), stride=2, padding=int((1 + k) / 2 - 1),
                               output_padding=1)
        return _gdn(spec['gdn'], y) if spec['nl'] in ('gdn', 'gdn_inverse') else _act(spec['nl'], y)
    if t == 'Conv2d':
        return _act(spec.get('nl'), F.conv2d(x, _t(spec['weight']), _t(spec.get('bias')), stride=spec['stride']))
    if t == 'ChengResBlock':
        if spec['mode'] == 'plain':
            return x + _run(spec['layers'], x)
        return _run(spec['aux'], x) + _run(spec['layers'], x)
    if t == 'ResBlock':
        h = F.relu(_conv(x, spec['w1'], spec['b1'], spec['k'], 1))
        return F.relu(x + _conv(h, spec['w2'], spec['b2'], spec['k'], 1))
    if t == 'AttentionResBlock':
        h = F.leaky_relu(F.conv2d(x, _t(spec['w1']), _t(spec['b1'])), 0.01)
        h = F.leaky_relu(_conv(h, spec['w2'], spec['b2'], 3, 1), 0.01)
        return F.leaky_relu(x + F.conv2d(h, _t(spec['w3']), _t(spec['b3'])), 0.01)
    if t == 'SimplifiedAttention':
        trunk = x
        for s in spec['trunk']:
            trunk = _run(s, trunk)
        att = x
        for s in spec['attention']:
            att = _run(s, att)
        return trunk * torch.sigmoid(F.conv2d(att, _t(spec['w_out']), _t(spec['b_out']))) + x
    raise ValueError('unknown layer spec type %r' % t)


def run_layer(spec, x, res=None, cmap=None):
    """oracle.run_layer's contract (NHWC numpy in / out; cmap is a layout hint of the fmaf kernels, unused here)"""
    with torch.no_grad():
        xt = torch.from_numpy(np.ascontiguousarray(x, np.float32)).permute(0, 3, 1, 2)
        y = _run(spec, xt)
        if res is not None:
            y = y + torch.from_numpy(np.ascontiguousarray(res, np.float32)).permute(0, 3, 1, 2)
        return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())


@contextlib.contextmanager
def torch_convs(threads):
    """inside: oracle.codec's transforms run on torch-CPU with `threads` intra-op threads"""
    keep, keep_threads = O.run_layer, torch.get_num_threads()
    torch.set_num_threads(int(threads))
    O.run_layer = run_layer
    try:
        yield
    finally:
        O.run_layer = keep
        torch.set_num_threads(keep_threads)
        _W.clear()
