"""GainMatrix (src/layers/multi_rate/gain_matrix.py:32-194): per-channel gain vectors per rate index,
|g|, with geometric interpolation for fractional idx_rate at inference."""
import numpy as np
import torch
from torch.nn import Module, Parameter, ParameterList

from ... import ops
from ...func_util.nn_util import get_value


class GainMatrix(Module):
    def __init__(self, param):
        super().__init__()
        default = {'N': None, 'nb_ft': None, 'initialize_to_one': True, 'scalar_gain': False}
        n = get_value('N', param, default)
        nb_ft = get_value('nb_ft', param, default)
        dim = (1, 1, 1) if get_value('scalar_gain', param, default) else (nb_ft, 1, 1)
        to_one = get_value('initialize_to_one', param, default)
        self.enc_gain_list = ParameterList()
        self.dec_gain_list = ParameterList()
        for _ in range(n):
            for lst in (self.enc_gain_list, self.dec_gain_list):
                g = torch.ones(dim) if to_one else torch.randn(dim) * (2.0 / int(np.prod(dim))) ** 0.5
                lst.append(Parameter(g))

    def get_gain(self, idx_rate, mode=''):
        lst = {'enc': self.enc_gain_list, 'dec': self.dec_gain_list}[mode]
        return lst[idx_rate].abs()

    def interpolate_gain_vector(self, idx_rate, mode=''):
        """g_r^l * g_t^(1-l) between the two neighbouring integer rate indices (tiny [C] vector:
        evaluated with torch on the parameter's device, once per call)."""
        prev_i = int(np.floor(idx_rate))
        next_i = prev_i + 1
        lam = 1 - (idx_rate - prev_i)
        if next_i == len(self.enc_gain_list):
            next_i = prev_i
        return (self.get_gain(prev_i, mode) ** lam) * (self.get_gain(next_i, mode) ** (1 - lam))

    def gain_vector(self, idx_rate, mode):
        """[C] gain used by the codec: the parameter itself for an integer rate index (|.| is applied
        by the kernels), else the deterministic interpolation kernel (aivc_gain_interp)."""
        if float(idx_rate) == int(idx_rate):
            return self.get_gain(int(idx_rate), mode).detach().reshape(-1)
        lst = {'enc': self.enc_gain_list, 'dec': self.dec_gain_list}[mode]
        prev_i = int(np.floor(idx_rate))
        next_i = prev_i + 1 if prev_i + 1 < len(lst) else prev_i
        lam = 1 - (idx_rate - prev_i)
        if not lst[prev_i].is_cuda:
            return self.interpolate_gain_vector(idx_rate, mode).detach().reshape(-1)  # API use on CPU params
        return ops.gain_interp(lst[prev_i], lst[next_i], lam)

    def forward(self, param):
        default = {'x': None, 'idx_rate': 0., 'mode': None}
        x = get_value('x', param, default)
        idx_rate = get_value('idx_rate', param, default)
        mode = get_value('mode', param, default)
        g = self.gain_vector(idx_rate, mode).to(x.device)
        if g.numel() == 1:
            g = g.expand(x.shape[1]).contiguous()
        y = ops.channel_gain(ops.to_nhwc(x), g)
        return {'output': ops.to_nchw_view(y)}
