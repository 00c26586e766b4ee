"""Module tree -> plain-data spec (nested dicts of numpy arrays) for the CPU oracle.  TEST
INFRASTRUCTURE ONLY.  Works by duck typing on class names, so it accepts this repo's modules and
the reference's own modules alike (same attribute layout)."""
import numpy as np


def _np(t):
    return None if t is None else t.detach().cpu().numpy().astype(np.float32)


def _nl_name(mod):
    if mod is None:
        return 'no'
    return {'LeakyReLU': 'leaky_relu', 'ReLU': 'relu', 'Sigmoid': 'sigmoid', 'GDN': 'gdn'}[type(mod).__name__]


def _gdn(m):
    return {'beta': _np(m.beta), 'gamma': _np(m.gamma), 'inverse': bool(m.inverse),
            'beta_bound': float(m.beta_bound), 'gamma_bound': float(m.gamma_bound), 'pedestal': float(m.pedestal)}


def export_spec(m):
    t = type(m).__name__
    if t == 'Sequential':
        return {'type': 'Sequential', 'layers': [export_spec(c) for c in m]}
    if t == 'CustomConvLayer':
        conv = m.layers[1]
        nl = m.layers._modules.get('non_linearity')
        d = {'type': t, 'k': conv.kernel_size[0], 'stride': conv.stride[0], 'weight': _np(conv.weight),
             'bias': _np(conv.bias), 'nl': _nl_name(nl)}
        if d['nl'] == 'gdn':
            d['gdn'] = _gdn(nl)
            d['nl'] = 'gdn_inverse' if nl.inverse else 'gdn'
        return d
    if t == 'UpscalingLayer':
        conv = m.layers[0]
        nl = m.layers._modules.get('non_linearity')
        d = {'type': t, 'k': conv.kernel_size[0], 'weight': _np(conv.weight), 'bias': _np(conv.bias),
             'nl': _nl_name(nl)}
        if d['nl'] == 'gdn':
            d['gdn'] = _gdn(nl)
            d['nl'] = 'gdn_inverse' if nl.inverse else 'gdn'
        return d
    if t == 'Conv2d':
        return {'type': 'Conv2d', 'weight': _np(m.weight), 'bias': _np(m.bias), 'stride': m.stride[0]}
    if t == 'ChengResBlock':
        d = {'type': t, 'mode': m.mode, 'layers': export_spec(m.layers)}
        if m.mode != 'plain':
            d['aux'] = export_spec(m.aux_layer)
        return d
    if t == 'ResBlock':
        c1, c2 = m.layers[1], m.layers[4]
        return {'type': t, 'k': c1.kernel_size[0], 'w1': _np(c1.weight), 'b1': _np(c1.bias), 'w2': _np(c2.weight),
                'b2': _np(c2.bias)}
    if t == 'AttentionResBlock':
        c1, c2, c3 = m.layers[0], m.layers[3], m.layers[5]
        return {'type': t, 'w1': _np(c1.weight), 'b1': _np(c1.bias), 'w2': _np(c2.weight), 'b2': _np(c2.bias),
                'w3': _np(c3.weight), 'b3': _np(c3.bias)}
    if t == 'SimplifiedAttention':
        att = list(m.attention)
        return {'type': t, 'trunk': [export_spec(c) for c in m.trunk], 'attention': [export_spec(c) for c in att[:3]],
                'w_out': _np(att[3].weight), 'b_out': _np(att[3].bias)}
    if t == 'GDN':
        return dict(_gdn(m), type='GDN')
    raise ValueError('export_spec: unsupported module %s' % t)


def export_balle(pe):
    c = pe.nb_channel
    parts = [_np(p).reshape(c, -1) for p in pe.matrix_h] + [_np(p).reshape(c, -1) for p in pe.bias_b] \
        + [_np(p).reshape(c, -1) for p in pe.bias_a]
    return np.ascontiguousarray(np.concatenate(parts, axis=1), np.float32)


def export_conditional(net):
    d = {k: export_spec(getattr(net, k)) for k in ('g_a', 'g_s', 'h_a', 'h_s')}
    d['g_a_ref'] = export_spec(net.g_a_ref) if getattr(net, 'g_a_ref', None) is not None else None
    d.update(c_y=net.nb_ft_y, c_z=net.nb_ft_z, c_short=net.out_c_shortcut_y, balle=export_balle(net.pdf_z),
             flag_gain_p_b=bool(net.flag_gain_p_b))
    gains = {'I': net.gain_I}
    if net.flag_gain_p_b:
        gains.update(P=net.gain_P, B=net.gain_B)
    d['gain'] = {k: {'enc': [np.abs(_np(g)).reshape(-1) for g in gm.enc_gain_list],
                     'dec': [np.abs(_np(g)).reshape(-1) for g in gm.dec_gain_list]} for k, gm in gains.items()}
    return d


def export_model(full_net):
    return {'mof': export_conditional(full_net.mode_net.mode_net),
            'cod': export_conditional(full_net.codec_net.codec_net)}
