"""Entropy-model modules with the reference's names (src/layers/entropy_coding/pdf_estimator.py).

BallePdfEstim holds the factorised-prior parameters (`matrix_h.{0..3}`, `bias_a.{0..2}`,
`bias_b.{0..3}`) exactly like the reference; what the codec needs from it -- the CDF at the 514
half-integer points -- is produced by the aivc_balle_cdf_table kernel.  ParametricPdf is kept for
pickle compatibility and rate estimation (logging: aivc_laplace_prob / aivc_table_prob, csrc/rate.hip)."""
import torch
from torch import nn

from ... import abi, ops
from .._cache import cached


def _xavier(shape):
    n = 1
    for s in shape:
        n *= s
    return torch.randn(shape) * (2.0 / n) ** 0.5


class ParametricPdf(nn.Module):
    def __init__(self, pdf_family):
        super().__init__()
        self.pdf_family = pdf_family

    def forward(self, y_tilde, all_pdf_param, zero_mu=False):
        """p(y_tilde) = sum over the mixture components of cdf(y + .5) - cdf(y - .5)
        (src/layers/entropy_coding/pdf_estimator.py:27-65); Laplace family (what the coded path uses:
        src/real_life/bitstream.py:127-154), evaluated by aivc_laplace_prob.  Rate estimation / logging only."""
        fam = self.pdf_family.split('_')
        if 'laplace' not in fam:
            raise NotImplementedError('ParametricPdf.forward: family %r (the coded path is Laplace)' % self.pdf_family)
        p = None
        for pdf_param in all_pdf_param:
            mu = None if ('mu' in fam or zero_mu) else pdf_param.get('mu')
            sigma = pdf_param.get('sigma').expand_as(y_tilde)
            cur = ops.laplace_prob(y_tilde, None if mu is None else mu.expand_as(y_tilde), sigma)
            p = cur if p is None else p + cur
        return p


class BallePdfEstim(nn.Module):
    """Factorised prior of Balle et al. 2018 (appendix 6): per-channel monotone MLP 1->3->3->3->1."""

    def __init__(self, nb_channel, pdf_family, verbose=True):
        super().__init__()
        self.nb_channel = nb_channel
        self.pdf_family = pdf_family
        self.K = 4
        self.r = 3
        self.matrix_h = nn.ParameterList()
        self.bias_b = nn.ParameterList()
        self.bias_a = nn.ParameterList()
        corr = float(nb_channel) ** 0.5
        dims = [1, self.r, self.r, self.r, 1]
        for i in range(self.K):
            self.matrix_h.append(nn.Parameter(_xavier((nb_channel, dims[i], dims[i + 1])) * corr))
            self.bias_b.append(nn.Parameter(_xavier((nb_channel, dims[i + 1])) * corr))
            if i != self.K - 1:
                self.bias_a.append(nn.Parameter(_xavier((nb_channel, dims[i + 1])) * corr))

    def packed_params(self, device):
        """[C][43] fp32 in the layout aivc_balle_cdf_table expects."""
        def build():
            c = self.nb_channel
            parts = [p.detach().reshape(c, -1) for p in self.matrix_h]
            parts += [p.detach().reshape(c, -1) for p in self.bias_b]
            parts += [p.detach().reshape(c, -1) for p in self.bias_a]
            out = torch.cat(parts, dim=1).to(device, torch.float32).contiguous()
            assert out.shape[1] == abi.BALLE_PARAMS
            return out
        params = list(self.matrix_h) + list(self.bias_b) + list(self.bias_a)
        return cached(self, ('balle', str(device)), params, build)

    def cdf_table(self, device, want_float=False):
        """uint16 CDF rows [C][CDF_ROW] (and optionally the fp32 CDF [C][514])."""
        if want_float:
            return ops.balle_cdf_table(self.packed_params(device), want_float=True)
        params = list(self.matrix_h) + list(self.bias_b) + list(self.bias_a)
        return cached(self, ('table', str(device)), params,
                      lambda: ops.balle_cdf_table(self.packed_params(device)))

    def cdf(self, x_tilde):
        """CDF at the reference's 514 evaluation points only ([1,C,514,1] -> [1,C,514,1])."""
        pts = torch.arange(abi.LP, device=x_tilde.device).float() - abi.AC_MAX_VAL - 0.5
        if tuple(x_tilde.shape[2:]) != (abi.LP, 1) or not torch.equal(x_tilde[0, 0, :, 0], pts):
            raise NotImplementedError('BallePdfEstim.cdf is only evaluated at k - 256.5, k = 0..513')
        _, cdf = self.cdf_table(x_tilde.device, want_float=True)
        return cdf.reshape(1, self.nb_channel, abi.LP, 1)

    def forward(self, x_tilde, pdf_param=None):
        """p(x_tilde) = cdf(x + .5) - cdf(x - .5) (src/layers/entropy_coding/pdf_estimator.py:185-202) for the
        integer-valued latents of inference, [B, C, H, W]: both points are entries of the 514-point table
        (aivc_table_prob); values that are not codable symbols give NaN.  Rate estimation / logging only."""
        if 'sigma' in self.pdf_family.split('_'):
            raise NotImplementedError('BallePdfEstim.forward: the sigma-scaled family is not used by the codec')
        _, cdf = self.cdf_table(x_tilde.device, want_float=True)
        return ops.table_prob(x_tilde, cdf)
