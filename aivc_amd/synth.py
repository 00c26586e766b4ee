"""Synthetic stand-ins for the assets the reference snapshot lacks: a seeded
moving-pattern YUV 4:2:0 clip and a randomly initialised model of the build's architecture."""
import math

import numpy as np
import torch

from .model_mngt.model_management import attach_arithmetic_coders
from .models import arch
from .models.full_net import FullNet


def synthetic_video(width, height, n_frames, seed=666, first=0, noise=4.0):
    """Planar 8-bit I420 frames: smooth translating pattern + noise.
    -> list of dicts {'y','u','v'} of uint8 numpy arrays."""
    rng = np.random.default_rng(seed)
    hc, wc = (height + 1) // 2, (width + 1) // 2
    xs, ys = np.arange(width)[None, :], np.arange(height)[:, None]
    xc, yc = np.arange(wc)[None, :], np.arange(hc)[:, None]
    out = []
    for t in range(first, first + n_frames):
        y = 128 + 64 * np.sin(2 * np.pi * (xs + 3 * t) / 97) + 48 * np.cos(2 * np.pi * (ys - 2 * t) / 61)
        u = 128 + 40 * np.sin(2 * np.pi * (xc + 1.5 * t) / 53) * np.cos(2 * np.pi * yc / 47)
        v = 128 + 40 * np.cos(2 * np.pi * (yc - t) / 41) * np.sin(2 * np.pi * xc / 59)
        f = {}
        for k, a, shp in (('y', y, (height, width)), ('u', u, (hc, wc)), ('v', v, (hc, wc))):
            a = a + rng.normal(0, noise, shp)
            f[k] = np.clip(np.rint(a), 0, 255).astype(np.uint8)
        out.append(f)
    return out


def to_device_frames(frames, device):
    return [{k: torch.from_numpy(f[k]).unsqueeze(0).to(device) for k in ('y', 'u', 'v')} for f in frames]


def _init_weights(model, gen):
    """Variance-preserving seeded init (the default torch init shrinks activations by ~3x per
    layer, which would make every latent round to zero and leave nothing to entropy-code)."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('.beta') or name.endswith('.gamma'):
                p.add_(torch.rand(p.shape, generator=gen) * 0.02)
            elif 'gain_list' in name:
                p.copy_(1.0 + 0.25 * torch.rand(p.shape, generator=gen))
            elif 'matrix_h' in name or 'bias_a' in name or 'bias_b' in name:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.8)
            elif p.dim() == 4:
                fan_in = p[0].numel() if 'layers.0.weight' not in name or p.shape[0] != p.shape[1] else p[0].numel()
                p.copy_(torch.randn(p.shape, generator=gen) * (1.3 / math.sqrt(fan_in)))
            elif p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)


def _scale_last(seq, scale, bias_shift=None):
    last = [m for m in seq.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))][-1]
    with torch.no_grad():
        last.weight.mul_(scale)
        if bias_shift is not None:
            last.bias.add_(bias_shift)


def make_model(widths=None, seed=1234, device=None):
    """Randomly initialised FullNet (eval mode, arithmetic coders attached)."""
    widths = widths or arch.DEFAULT_WIDTHS
    torch.manual_seed(seed)
    model = FullNet({'widths': widths})
    gen = torch.Generator().manual_seed(seed)
    _init_weights(model, gen)
    for net in (model.mode_net.mode_net, model.codec_net.codec_net):
        c_y = net.nb_ft_y
        _scale_last(net.g_a, 6.0)       # latents with a few levels of dynamic
        _scale_last(net.g_a_ref, 3.0)
        shift = torch.zeros(2 * c_y)
        shift[c_y:] = 0.5               # log-variance bias -> sigma around 1.3
        _scale_last(net.h_s, 1.0, shift)
        _scale_last(net.h_a, 3.0)
    _scale_last(model.mode_net.mode_net.g_s, 0.6)   # flows of a few pixels, alpha/beta mid-range
    _scale_last(model.codec_net.codec_net.g_s, 0.15, torch.full((3,), 0.45))
    model = model.eval()
    if device is not None:
        model = model.to(device)
    return attach_arithmetic_coders(model)


def calibrate_operating_point(model, device, active_y=(6, 12), target_std=0.8, size=(144, 256), seed=5):
    """Shape the latent statistics of a random-init model into a plausible low-rate operating point
    (a trained ms_ssim-4 model sends a minority of its y feature maps, the others are all-zero and
    skipped by the bitstream; README.md:25-26 quotes 1-20 Mbit/s at 1080p across the 7 models).

    Runs g_a once per net on a small synthetic clip ON THE GPU (HIP path), then rescales the last
    analysis conv per output channel: the first `active_y[i]` channels get standard deviation
    `target_std` (zero mean), the others are made negligible so they quantise to all-zero maps.
    The hyper-synthesis is biased towards mu ~ 0, sigma ~ target_std."""
    from . import ops
    from .models.conditional_net import run_nhwc
    h, w = size
    frames = to_device_frames(synthetic_video(w, h, 3, seed=seed), device)
    f444 = [ops.yuv420_to_444(f['y'], f['u'], f['v'], c_store=3) for f in frames]
    inputs = {'mof': torch.cat((f444[1], f444[0], f444[2]), dim=3),
              'cod': torch.cat((f444[1], 0.5 * f444[0]), dim=3)}
    nets = {'mof': model.mode_net.mode_net, 'cod': model.codec_net.codec_net}
    with torch.no_grad():
        for (name, net), n_act in zip(nets.items(), active_y):
            y = run_nhwc(net.g_a, inputs[name]).reshape(-1, net.nb_ft_y)
            # statistics with exactly rounded sums (math.fsum) on the host: the encode and decode CLIs rebuild this
            # model in separate processes, possibly on different GPUs -- a reduction whose order depends on the
            # device or the torch build could move a weight by an ulp and desynchronise the entropy decoder.
            # y itself is bit-reproducible (fixed-order HIP kernels).
            cols = y.double().cpu().numpy().T
            n_s = cols.shape[1]
            m64 = [math.fsum(c) / n_s for c in cols]
            s64 = [max(math.sqrt(math.fsum((v - m) * (v - m) for v in c) / (n_s - 1)), 1e-6) for c, m in zip(cols, m64)]
            mean = torch.tensor(m64, dtype=torch.float64).float().to(device)
            std = torch.tensor(s64, dtype=torch.float64).float().to(device)
            scale = torch.full_like(std, 0.02) / std
            scale[:n_act] = target_std / std[:n_act]
            last = [m for m in net.g_a.modules() if isinstance(m, torch.nn.Conv2d)][-1]
            last.weight.mul_(scale.view(-1, 1, 1, 1).to(last.weight.device))
            last.bias.copy_(((last.bias.to(device) - mean) * scale).to(last.bias.device))
            hs_last = [m for m in net.h_s.modules() if isinstance(m, torch.nn.Conv2d)][-1]
            c = net.nb_ft_y
            hs_last.weight[:c].mul_(0.05)
            hs_last.bias[:c].zero_()
            hs_last.weight[c:].mul_(0.3)
            hs_last.bias[c:].fill_(2.0 * math.log(target_std))
    return model
