"""MotionCompensation: x_warp = beta * warp(prev, v_prev) + (1 - beta) * warp(next, v_next).
Build-authored stand-in for the module missing from the snapshot; call contract from
src/real_life/decode.py:524-533 (dict in, {'x_warp': ...} out)."""
from torch.nn import Module

from ..func_util.nn_util import get_value
from ..func_util.optical_flow import warp


class MotionCompensation(Module):
    def forward(self, param):
        default = {'prev': None, 'next': None, 'v_prev': None, 'v_next': None, 'beta': None,
                   'interpol_mode': 'bilinear'}
        prev, nxt = get_value('prev', param, default), get_value('next', param, default)
        v_prev, v_next = get_value('v_prev', param, default), get_value('v_next', param, default)
        beta = get_value('beta', param, default)
        mode = get_value('interpol_mode', param, default)
        # stand-alone API path (the codec itself uses the fused aivc_warp_blend kernel)
        wp, wn = warp(prev, v_prev, interpol_mode=mode), warp(nxt, v_next, interpol_mode=mode)
        return {'x_warp': beta * wp + (1 - beta) * wn}
