"""GDN / Quantizer / PdfParamParameterizer / LowerBound / View with the reference's class and
attribute names (src/layers/misc/misc_layers.py), computing through the HIP library.

Inference only: the modules expose the parameters and buffers a reference pickle carries, their
forward() runs the gfx950 kernels (aivc_gdn_reparam + aivc_conv2d in GDN mode, aivc_quantize_center,
aivc_hyper_params).  CPU tensors are rejected (the CPU restatement is oracle/, tests only)."""
import torch
from torch import nn
from torch.autograd import Function

from ... import ops
from .._cache import cached

LOG_VAR_MIN, LOG_VAR_MAX = -18.4207, 10.0  # reference: src/func_util/math_func.py:26-31


class View(nn.Module):
    def __init__(self, shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x.view(*self.shape)


class LowerBound(Function):
    """max(inputs, bound) (forward only; training is out of scope, src/layers/misc/misc_layers.py:39-60)."""

    @staticmethod
    def forward(ctx, inputs, bound):
        return torch.clamp(inputs, min=float(bound))

    @staticmethod
    def backward(ctx, grad_output):
        raise NotImplementedError('aivc_amd is inference only')


class GDN(nn.Module):
    """y_i = x_i / sqrt(beta_i + sum_j gamma_ij x_j^2)  (inverse: multiply).
    Reference: src/layers/misc/misc_layers.py:63-154.  Parameters `beta` [C], `gamma` [C, C]; plain
    tensor attributes `pedestal`, `beta_bound`, `gamma_bound`, `reparam_offset` travel in pickles."""

    def __init__(self, ch, inverse=False, beta_min=1e-6, gamma_init=.1, reparam_offset=2 ** -18):
        super().__init__()
        self.inverse = inverse
        self.beta_min = beta_min
        self.gamma_init = gamma_init
        self.reparam_offset = torch.FloatTensor([reparam_offset])
        self.current_device = 'cpu'
        self.pedestal = self.reparam_offset ** 2
        self.beta_bound = (self.beta_min + self.reparam_offset ** 2) ** .5
        self.gamma_bound = self.reparam_offset
        self.beta = nn.Parameter(torch.sqrt(torch.ones(ch) + self.pedestal))
        self.gamma = nn.Parameter(torch.sqrt(self.gamma_init * torch.eye(ch) + self.pedestal))

    def effective_params(self, device):
        """(beta_eff, gamma_eff) on `device`, re-parameterised once and cached."""
        def build():
            beta = self.beta.detach().to(device, torch.float32)
            gamma = self.gamma.detach().to(device, torch.float32)
            return ops.gdn_reparam(beta, gamma, float(self.beta_bound), float(self.gamma_bound),
                                   float(self.pedestal))
        return cached(self, ('gdn', str(device)), (self.beta, self.gamma), build)

    def forward_nhwc(self, x, res=None):
        be, ge = self.effective_params(x.device)
        return ops.gdn(x, be, ge, inverse=self.inverse, res=res)

    def forward(self, inputs):
        if inputs.dim() == 5:  # the reference folds a 5-D tensor to 4-D and back
            bs, ch, d, w, h = inputs.size()
            out = self.forward(inputs.reshape(bs, ch, d * w, h))
            return out.reshape(bs, ch, d, w, h)
        return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(inputs)))


class Quantizer(nn.Module):
    """Inference quantiser: round half to even (src/layers/misc/misc_layers.py:157-169)."""

    def forward(self, x, fine_tune=False):
        if self.training or fine_tune:
            raise NotImplementedError('aivc_amd is inference only (no additive-noise training path)')
        _, y = ops.quantize_center(x.contiguous().view(1, 1, -1, 1))
        return y.view(x.shape)


class PdfParamParameterizer(nn.Module):
    """Splits h_s output into mu / sigma = exp(0.5 clamp(logvar)) (K = 1 Laplace case used by the
    codec; src/layers/misc/misc_layers.py:172-269)."""

    def __init__(self, ec_mode, nb_ft):
        super().__init__()
        self.ec_mode = ec_mode
        self.nb_ft = nb_ft

    def forward_nhwc(self, x, h=None, w=None):
        n, hh, wh, _ = x.shape
        return ops.hyper_params(x, self.nb_ft, hh if h is None else h, wh if w is None else w)

    def forward(self, x):
        toks = self.ec_mode.split('_')
        if 'two' in toks or 'three' in toks or 'gamma' in toks:
            raise NotImplementedError('mixture entropy models are not part of the coded path')
        mu, sigma = self.forward_nhwc(ops.to_nhwc(x))
        mu, sigma = ops.to_nchw_view(mu), ops.to_nchw_view(sigma)
        return [{'mu': mu, 'sigma': sigma, 'gamma': torch.ones_like(mu), 'weight': torch.ones_like(mu)}]
