"""Simplified attention module of Cheng et al. 2020 with the reference's names
(src/layers/misc/attention.py:22-97).  The whole `trunk * sigmoid(att) + x` tail is the epilogue of
the last 1x1 convolution (sigmoid, multiply and residual add fused in aivc_conv2d)."""
from torch import nn
from torch.nn import Conv2d, LeakyReLU, ReplicationPad2d, Sequential, Sigmoid

from ... import abi, ops
from .custom_conv_layers import ResBlock, run_conv


class AttentionResBlock(nn.Module):
    """leaky(x + conv1x1(leaky(conv3x3(leaky(conv1x1(x)))))) at nb_ft/2 inner width."""

    def __init__(self, nb_ft):
        super().__init__()
        half = int(nb_ft / 2)
        self.layers = Sequential(Conv2d(nb_ft, half, 1), LeakyReLU(), ReplicationPad2d(1),
                                 Conv2d(half, half, 3), LeakyReLU(), Conv2d(half, nb_ft, 1))

    def forward_nhwc(self, x):
        h = run_conv(self.layers[0], x, 0, act1=abi.ACT_LEAKY)
        # 3x3 conv and the closing 1x1 in one launch: the nb_ft/2-wide intermediate stays on the chip
        return run_conv(self.layers[3], h, 1, act1=abi.ACT_LEAKY, tail=self.layers[5], res=x, act2=abi.ACT_LEAKY)

    def forward(self, x):
        return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(x)))


class SimplifiedAttention(nn.Module):
    def __init__(self, nb_ft, k_size=3, lightweight_resblock=False):
        super().__init__()
        self.nb_ft = nb_ft
        self.k_size = k_size

        def block():
            return AttentionResBlock(nb_ft) if lightweight_resblock else ResBlock(k_size, nb_ft)
        self.trunk = Sequential(block(), block(), block())
        self.attention = Sequential(block(), block(), block(), Conv2d(nb_ft, nb_ft, 1), Sigmoid())

    def forward_nhwc(self, x):
        t = x
        for m in self.trunk:
            t = m.forward_nhwc(t)
        a = x
        for m in list(self.attention)[:3]:
            a = m.forward_nhwc(a)
        return run_conv(self.attention[3], a, 0, act1=abi.ACT_SIGMOID, mul=t, res=x)

    def forward(self, x):
        return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(x)))
