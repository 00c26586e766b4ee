#!/usr/bin/env python3
"""How often does a stream written by ANOTHER implementation of the transforms desynchronise this decoder?
(build container only: imports the reference's layer classes from /root/reference)

The y sections are coded with CDFs built from sigma = exp(0.5 * clamp(h_s(z_hat))).  The reference evaluates h_s with
ATen / oneDNN convolutions, this build with fixed-order fmaf chains (HIP == oracle, bit for bit): the two sigmas differ in
their last bits, now and then a 16-bit CDF entry flips, and a flipped bound of a CODED symbol takes an arithmetic decoder
off the writer's track for the rest of the section.  This script measures, at the DEFAULT widths (the bench's synthetic
h_s: 32 -> 128 -> 128 -> 128 channels, 1080p latent sizes 17x30 -> 68x120):

  * the relative difference of the two sigmas,
  * the fraction of the 514 CDF entries per position that differ,
  * the fraction of coded symbols (q ~ round(Laplace(0, sigma / sqrt 2)), the model's own statistics) with a differing bound,

and derives P(desync) per frame at 1080p for the bench's operating point (6 + 12 coded maps) and for every map coded.
The same CDF arithmetic is used on both sigmas (the reference's torch Laplace.cdf + torchac normalisation and this
build's agree on every entry given the same sigma: `cdf_stats` of the decoder fixtures, 0 of 11 M).

    python tools/desync_stats.py [--draws 4] > profiles/r04_desync_stats.json
"""
import argparse
import json
import os
import sys
import types

sys.dont_write_bytecode = True  # the reference tree is read-only input: importing it must not leave __pycache__ there

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
REF = '/root/reference/src'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--draws', type=int, default=4)
    ap.add_argument('--threads', type=int, default=16)
    a = ap.parse_args()
    from oracle import oracle as O
    from oracle import spec as ospec
    from aivc_amd import synth
    from aivc_amd.models import arch
    O.build()
    torch.set_num_threads(a.threads)
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device='cpu')
    net = model.codec_net.codec_net
    hs_spec = ospec.export_spec(net.h_s)
    # the reference's h_s with the same parameters
    tv = types.ModuleType('torchvision'); tvt = types.ModuleType('torchvision.transforms'); tvf = types.ModuleType('torchvision.transforms.functional')
    tvf.to_tensor = tvf.to_pil_image = lambda *x, **k: None
    tv.transforms, tvt.functional = tvt, tvf
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tvt, 'torchvision.transforms.functional': tvf,
                        'torchac': types.ModuleType('torchac')})
    sys.path.insert(0, REF)
    import func_util.console_display as cd
    cd.FLAG_QUIET = True
    from layers.misc.custom_conv_layers import CustomConvLayer, UpscalingLayer
    from layers.misc.misc_layers import PdfParamParameterizer
    wd = arch.DEFAULT_WIDTHS
    ref_hs = torch.nn.Sequential(UpscalingLayer(5, wd['c_z'], wd['n_h'], non_linearity='leaky_relu'),
                                 UpscalingLayer(5, wd['n_h'], wd['n_h'], non_linearity='leaky_relu'),
                                 CustomConvLayer(3, wd['n_h'], 2 * wd['c_y'], non_linearity='no')).eval()
    ref_hs.load_state_dict(net.h_s.state_dict(), strict=True)
    ref_pp = PdfParamParameterizer('laplace', wd['c_y'])
    h_y, w_y, h_z, w_z, c_y = 68, 120, 17, 30, wd['c_y']
    rng = np.random.default_rng(2024)
    tot_entries = flip_entries = tot_sym = bad_sym = 0
    rel_max, rel_sum, n_sigma = 0.0, 0.0, 0
    for d in range(a.draws):
        z = np.rint(rng.laplace(0, 1.5, (1, h_z, w_z, wd['c_z']))).astype(np.float32)
        with torch.no_grad():
            p = ref_pp(ref_hs(torch.from_numpy(z).permute(0, 3, 1, 2))[:, :, :h_y, :w_y])
        sig_t = np.ascontiguousarray(p[0]['sigma'].permute(0, 2, 3, 1).numpy())
        _, sig_o = O.hyper_params(O.run_layer(hs_spec, z), c_y, h_y, w_y)
        rel = np.abs(sig_o / sig_t - 1)
        rel_max, rel_sum, n_sigma = max(rel_max, float(rel.max())), rel_sum + float(rel.sum()), n_sigma + rel.size
        q = np.clip(np.rint(rng.laplace(0, 1, sig_t.shape) * sig_t / np.sqrt(2)), -256, 255).astype(np.int64)
        for m0 in range(0, c_y, 8):
            maps = list(range(m0, m0 + 8))
            rt = O.laplace_cdf_rows(sig_t, maps)[:, :514]
            ro = O.laplace_cdf_rows(sig_o, maps)[:, :514]
            diff = rt != ro
            tot_entries += diff.size
            flip_entries += int(diff.sum())
            sym = np.concatenate([q[0, :, :, m].reshape(-1) for m in maps]) + 256  # rows are map-major, then positions
            ar = np.arange(len(sym))
            bad = diff[ar, sym] | diff[ar, sym + 1]
            tot_sym += len(sym)
            bad_sym += int(bad.sum())
        sys.stderr.write('draw %d: entries %d flipped %d, coded symbols %d with a differing bound %d\n'
                         % (d, tot_entries, flip_entries, tot_sym, bad_sym))
    p_sym = bad_sym / tot_sym
    npos = h_y * w_y

    def p_frame(n_maps):
        return 1.0 - (1.0 - p_sym) ** (n_maps * npos)
    out = {'setup': 'default widths (h_s 32 -> 128 -> 128 -> 2 x 64), 1080p latents (z 17 x 30 -> y 68 x 120), bench model seed 1234, '
                    'z_hat ~ round(Laplace(0, 1.5)), %d draws; reference h_s on torch %s CPU (%d threads) vs the oracle (== HIP)'
                    % (a.draws, torch.__version__, a.threads),
           'sigma_rel_diff': {'max': rel_max, 'mean': rel_sum / n_sigma, 'values': n_sigma},
           'cdf_entries': {'compared': tot_entries, 'differing': flip_entries, 'rate': flip_entries / tot_entries},
           'coded_symbols': {'compared': tot_sym, 'with_a_differing_bound': bad_sym, 'rate': p_sym},
           'p_desync_per_1080p_frame': {'I frame, 12 coded maps (bench operating point)': p_frame(12),
                                        'P/B frame, 6 + 12 coded maps (bench operating point)': p_frame(18),
                                        'P/B frame, every map coded (64 + 64)': p_frame(128)},
           'expected_coded_symbols_until_first_desync': (1.0 / p_sym) if p_sym else None}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
