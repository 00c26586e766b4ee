"""CodecNet wrapper: `.codec_net` is the ConditionalNet (src/model_mngt/model_management.py:351-354)."""
from torch.nn import Module

from .conditional_net import ConditionalNet


class CodecNet(Module):
    def __init__(self, param):
        super().__init__()
        p = dict(param)
        p.update({'in_c': 6, 'in_c_shortcut': 3, 'out_c': 3})  # (code || alpha*pred), alpha*pred -> x_hat
        self.codec_net = ConditionalNet(p)
