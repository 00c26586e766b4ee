"""ConditionalNet: one conditional coder (analysis g_a, conditioning transform g_a_ref, synthesis g_s,
hyperprior h_a / h_s, factorised prior on z, Laplace on y, per-frame-type gain matrices).

Authored by this build: the reference's class is missing from the snapshot; every
attribute the reference's decoder reads is present under the same name
(src/real_life/decode.py:770-795): g_s, h_s, g_a_ref, pdf_y, pdf_z, pdf_parameterizer,
out_c_shortcut_y, nb_ft_y, nb_ft_z, gain_I, flag_gain_p_b, gain_P, gain_B, ac.
The encoder side mirrors the decoder dataflow of src/real_life/decode.py:798-898.
"""
import torch
from torch.nn import Module

from .. import ops
from ..func_util.GOP_structure import FRAME_B, FRAME_I, FRAME_P
from ..func_util.nn_util import get_value
from ..layers.entropy_coding.entropy_coder import EntropyCoder
from ..layers.entropy_coding.pdf_estimator import BallePdfEstim, ParametricPdf
from ..layers.misc.misc_layers import PdfParamParameterizer, Quantizer
from ..layers.multi_rate.gain_matrix import GainMatrix
from . import arch


def run_nhwc(transform, x):
    """Apply a Sequential of this package's layers on an NHWC tensor (no layout round trips)."""
    for m in transform:
        x = m.forward_nhwc(x)
    return x


class ConditionalNet(Module):
    def __init__(self, param):
        super().__init__()
        default = {'in_c': 3, 'in_c_shortcut': 3, 'out_c': 3, 'widths': arch.DEFAULT_WIDTHS,
                   'nb_rates': 1, 'flag_gain_p_b': True, 'flag_g_a_ref': True}
        in_c = get_value('in_c', param, default)
        in_c_shortcut = get_value('in_c_shortcut', param, default)
        out_c = get_value('out_c', param, default)
        wd = get_value('widths', param, default)
        nb_rates = get_value('nb_rates', param, default)
        self.nb_ft_y = wd['c_y']
        self.nb_ft_z = wd['c_z']
        self.out_c_shortcut_y = wd['c_short']
        self.g_a = arch.analysis_transform(in_c, self.nb_ft_y, wd)
        # "Some models don't have the shortcut transform" (src/real_life/decode.py:772-776): g_s then sees zeros
        self.g_a_ref = arch.analysis_transform(in_c_shortcut, self.out_c_shortcut_y, wd) \
            if get_value('flag_g_a_ref', param, default) else None
        self.g_s = arch.synthesis_transform(self.nb_ft_y + self.out_c_shortcut_y, out_c, wd)
        self.h_a = arch.hyper_analysis(self.nb_ft_y, self.nb_ft_z, wd)
        self.h_s = arch.hyper_synthesis(self.nb_ft_z, self.nb_ft_y, wd)
        self.pdf_y = ParametricPdf('laplace')
        self.pdf_z = BallePdfEstim(self.nb_ft_z, 'balle', verbose=False)
        self.pdf_parameterizer = PdfParamParameterizer('laplace', self.nb_ft_y)
        self.quantizer = Quantizer()
        self.entropy_coder = EntropyCoder()
        self.flag_gain_p_b = get_value('flag_gain_p_b', param, default)
        gm = {'N': nb_rates, 'nb_ft': self.nb_ft_y, 'initialize_to_one': True}
        self.gain_I = GainMatrix(gm)
        if self.flag_gain_p_b:
            self.gain_P = GainMatrix(gm)
            self.gain_B = GainMatrix(gm)
        self.ac = None  # ArithmeticCoder, attached by model_mngt.load_model (not pickled upstream)

    # ------------------------------------------------------------------------------------------
    def gain_module(self, frame_type):
        if not self.flag_gain_p_b or frame_type == FRAME_I:
            return self.gain_I
        return self.gain_P if frame_type == FRAME_P else self.gain_B

    def shortcut(self, in_shortcut, n, h_y, w_y, device, bands=None):
        """g_a_ref(in_shortcut) or the all-zero dummy (src/real_life/decode.py:887-892)."""
        if in_shortcut is not None and getattr(self, 'g_a_ref', None) is not None:
            s = run_nhwc(self.g_a_ref, in_shortcut)
            if bands is not None:  # computed in row bands over the ranks of the group: every rank gets the whole latent
                s = bands.gather_full(s)
            return s[:, :h_y, :w_y, :].contiguous() if s.shape[1:3] != (h_y, w_y) else s
        return torch.zeros((n, h_y, w_y, self.out_c_shortcut_y), dtype=torch.float32, device=device)

    def analyse(self, x_in, frame_type, idx_rate=0., bands=None):
        """Encoder side up to the quantised latents.  x_in NHWC.  Returns a dict with q_z / q_y
        (int16 NHWC), sigma, mu, y_hat (already multiplied by the decoder gain) and the latent sizes.
        bands (aivc_amd.bands.BandCtx): x_in is a banded input; g_a runs in row bands over the ranks of the group, y is
        all-gathered and the hyperprior + quantisation run on every rank alike (identical results: fixed-order
        arithmetic)."""
        gm = self.gain_module(frame_type)
        dev = x_in.device
        y = run_nhwc(self.g_a, x_in)
        if bands is not None:
            y = bands.gather_full(y)
        y = ops.channel_gain(y, gm.gain_vector(idx_rate, 'enc').to(dev))
        z = run_nhwc(self.h_a, y)
        q_z, z_hat = ops.quantize_center(z)
        n, h_y, w_y, _ = y.shape
        mu, sigma = ops.hyper_params(run_nhwc(self.h_s, z_hat), self.nb_ft_y, h_y, w_y)
        q_y, y_hat = ops.quantize_center(y, mu, gm.gain_vector(idx_rate, 'dec').to(dev))
        return {'q_z': q_z, 'q_y': q_y, 'mu': mu, 'sigma': sigma, 'y_hat': y_hat,
                'dim_y': (h_y, w_y), 'dim_z': tuple(z.shape[1:3])}

    def latents_from_symbols(self, q_z, q_y_fn, frame_type, dim_y, idx_rate=0.):
        """Decoder side: q_z (int16 NHWC) -> (mu, sigma); q_y_fn(sigma) must return q_y; -> y_hat."""
        gm = self.gain_module(frame_type)
        z_hat = ops.dequantize(q_z)
        mu, sigma = ops.hyper_params(run_nhwc(self.h_s, z_hat), self.nb_ft_y, dim_y[0], dim_y[1])
        q_y = q_y_fn(sigma)
        return ops.dequantize(q_y, mu, gm.gain_vector(idx_rate, 'dec').to(q_z.device))

    def synthesise(self, y_hat, in_shortcut, bands=None):
        """bands: g_a_ref and g_s run in row bands; -> this rank's band of the synthesis output"""
        n, h_y, w_y, _ = y_hat.shape
        s = self.shortcut(in_shortcut, n, h_y, w_y, y_hat.device, bands)
        x = torch.cat((y_hat, s), dim=3)
        return run_nhwc(self.g_s, x if bands is None else bands.full(x, 0))
