"""InputLayer / OutputLayer (src/layers/ae/ae_layers.py:17-56) on the HIP frame kernels."""
from torch.nn import Module

from ... import ops


def _plane(t):
    return t.reshape(t.shape[0], t.shape[-2], t.shape[-1])


class InputLayer(Module):
    """YUV 4:2:0 dict {'y','u','v'} of [B,1,H,W] tensors -> [B,3,H,W] (nearest x2 chroma, crop)."""

    def forward(self, x):
        y, u, v = x.get('y'), x.get('u'), x.get('v')
        out = ops.yuv420_to_444(_plane(y), _plane(u), _plane(v), c_store=3)
        return ops.to_nchw_view(out)


class OutputLayer(Module):
    """[B,3,H,W] -> YUV dict; U,V by 2x2 mean (bilinear x0.5, align_corners=False), floor size.
    Values are NOT cast to 8-bit levels here (the reference casts later, decode.py:575)."""

    def __init__(self, k_size=5):
        super().__init__()

    def forward(self, x):
        uv = ops.downsample2x(ops.to_nhwc(x), 1, 2)
        return {'y': x[:, 0:1, :, :], 'u': uv[:, 0:1], 'v': uv[:, 1:2]}
