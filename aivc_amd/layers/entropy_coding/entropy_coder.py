"""EntropyCoder (src/layers/entropy_coding/entropy_coder.py:18-30): rate estimate, logging only."""
from torch import nn


class EntropyCoder(nn.Module):
    def forward(self, prob_x, x):
        raise NotImplementedError('rate estimation is not part of the encode/decode hot path')
