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
