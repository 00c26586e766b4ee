"""Image helpers with the reference's names (src/func_util/img_processing.py)."""
import torch

from .. import ops
from .nn_util import get_value


def get_y_u_v(x):
    return x.get('y'), x.get('u'), x.get('v')


def cast_before_png_saving(param):
    """round(255 * clamp(x, 0, 1)) / 255, half to even (src/func_util/img_processing.py:31-75)."""
    default = {'x': None, 'data_type': 'yuv_dic'}
    x = get_value('x', param, default)
    data_type = get_value('data_type', param, default)

    def cast(t):
        flat = t.contiguous().view(1, 1, -1, 1)
        # x*1 through the 3-channel reconstruction kernel would also work; a 1-channel image cast is
        # the quantise-to-8-bit special case of aivc_frame_to_yuv420 with h = numel, w = 1
        pad = torch.zeros((1, flat.shape[2], 1, 3), dtype=torch.float32, device=t.device)
        pad[..., 0] = flat[..., 0]
        (y, _, _), _ = ops.frame_to_yuv420(pad, flat.shape[2], 1, want_u8=False)
        return y.view(t.shape)
    if data_type == 'tensor':
        return cast(x)
    return {c: cast(x.get(c)) for c in ('y', 'u', 'v')}


def interpolate_nearest(x, scale=2):
    return torch.nn.functional.interpolate(x, scale_factor=scale, mode='nearest')
