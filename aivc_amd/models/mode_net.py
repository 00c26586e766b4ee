"""ModeNet (MOFNet) wrapper: `.mode_net` is the ConditionalNet (src/model_mngt/model_management.py:356-359)."""
from torch.nn import Module

from .conditional_net import ConditionalNet


class ModeNet(Module):
    def __init__(self, param):
        super().__init__()
        p = dict(param)
        p.update({'in_c': 9, 'in_c_shortcut': 6, 'out_c': 6})  # (code||prev||next), (prev||next) -> alpha,beta,v_prev,v_next
        self.mode_net = ConditionalNet(p)
