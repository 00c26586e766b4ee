"""Model variants of the decoder fixtures (tools/gen_golden_decoder.py), derived from a stored base model by edits
that work on the reference's module tree and on this repo's alike (same attribute layout): the generator applies
them to the reference-built model before it writes the streams, the tests apply them to aivc_amd's FullNet after
loading the same weights -- so a variant costs no second set of weights in tests/golden/."""
import torch


def _last_conv(seq):
    return [m for m in seq.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))][-1]


def apply_variant(model, variant):
    """variant keys:
      drop_g_a_ref   -- "Some models don't have the shortcut transform" (src/real_life/decode.py:772-776)
      drop_gain_p_b  -- flag_gain_p_b False: gain_I for every frame type (decode.py:874-885)
      mof_active_y 0 -- MOFNet's y is identically zero: EMPTY y sections (src/real_life/bitstream.py:265-266, 430-466)"""
    nets = (model.mode_net.mode_net, model.codec_net.codec_net)
    with torch.no_grad():
        if variant.get('drop_g_a_ref'):
            for net in nets:
                net.g_a_ref = None
        if variant.get('drop_gain_p_b'):
            for net in nets:
                net.flag_gain_p_b = False
                for k in ('gain_P', 'gain_B'):
                    if k in net._modules:
                        del net._modules[k]
        if variant.get('mof_active_y') == 0:
            net = nets[0]
            c = net.nb_ft_y
            ga, hs = _last_conv(net.g_a), _last_conv(net.h_s)
            ga.weight.zero_()  # y == 0 ...
            ga.bias.zero_()
            hs.weight[:c].zero_()  # ... and mu == 0: round(y - mu) == 0 on every map
            hs.bias[:c].zero_()
    return model


def seeded_init(model, seed, active_y, weight_grid=None):
    """Initialise a FullNet look-alike (the reference-built one of tools/gen_golden_decoder.py or aivc_amd's: same
    attribute layout and parameter names) from numpy's frozen legacy generator, so that a fixture can carry the seed
    and a digest instead of megabytes of weights (the mid-width model has 3.2 M parameters).  Parameters are drawn in
    sorted-name order; the operating point is then shaped as for the stored models (only `active_y` maps of each
    network's y are non-zero; latents stay inside the coder's alphabet; the output lands in the 8-bit range).
    -> sha256 over every state_dict tensor."""
    import hashlib
    import math

    import numpy as np
    rs = np.random.RandomState(seed)
    params = dict(model.named_parameters())
    with torch.no_grad():
        for name in sorted(params):
            p = params[name]
            shape = tuple(p.shape)
            if name.endswith('.beta') or name.endswith('.gamma'):
                a = p.detach().numpy().astype(np.float64) + rs.uniform(0., .02, shape)
            elif 'gain_list' in name:
                a = 0.8 + 0.5 * rs.uniform(0., 1., shape)
            elif 'matrix_h' in name or 'bias_a' in name or 'bias_b' in name:
                a = rs.standard_normal(shape) * 0.8
            elif p.dim() == 4:
                a = rs.standard_normal(shape) / math.sqrt(int(np.prod(shape[1:])))
            elif p.dim() == 1:
                a = rs.standard_normal(shape) * 0.05
            else:
                continue
            p.copy_(torch.from_numpy(np.ascontiguousarray(a, np.float32)))
        for net, n_act in ((model.mode_net.mode_net, active_y[0]), (model.codec_net.codec_net, active_y[1])):
            c = net.nb_ft_y
            ga, hs, ha = _last_conv(net.g_a), _last_conv(net.h_s), _last_conv(net.h_a)
            ga.weight.mul_(5.0)
            ga.weight[n_act:].zero_()  # y == 0 and mu == 0 on the inactive maps: skipped by the bitstream
            ga.bias[n_act:].zero_()
            hs.weight[n_act:c].zero_()
            hs.bias[n_act:c].zero_()
            hs.weight[:c].mul_(0.3)
            hs.bias[c:].add_(0.3)
            ha.weight.mul_(3.0)
        _last_conv(model.mode_net.mode_net.g_s).weight.mul_(0.6)
        cg = _last_conv(model.codec_net.codec_net.g_s)
        cg.weight.mul_(0.15)
        cg.bias.add_(0.45)
        if weight_grid:
            for p in model.parameters():
                if p.dim() == 4:
                    p.copy_(torch.round(p * weight_grid) / weight_grid)
    h = hashlib.sha256()
    sd = model.state_dict()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()
