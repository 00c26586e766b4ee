"""EntropyCoder (src/layers/entropy_coding/entropy_coder.py:18-30): the rate a probability costs, -log2(p) with p
clamped to [PROBA_MIN, PROBA_MAX].  Logging / rate estimation only -- the coded path's rate is the bitstream size --
evaluated by aivc_rate_bits (csrc/rate.hip)."""
from torch import nn

from ... import ops
from ...func_util.math_func import PROBA_MAX, PROBA_MIN


class EntropyCoder(nn.Module):
    def forward(self, prob_x, x):
        rate, _ = ops.rate_bits(prob_x, PROBA_MIN, PROBA_MAX)
        return rate
