"""warp() with the reference's signature (src/func_util/optical_flow.py:14-55) on the HIP kernel."""
from .. import ops


def warp(x, flo, interpol_mode='bilinear', padding_mode='border', align_corners=True):
    """x [B,C,H,W], flo [B,2,H,W] (pixel units, channel 0 horizontal) -> warped [B,C,H,W]."""
    if interpol_mode != 'bilinear' or padding_mode != 'border' or not align_corners:
        raise NotImplementedError('only bilinear / border / align_corners=True (the codec\'s mode)')
    return ops.to_nchw_view(ops.warp(ops.to_nhwc(x), ops.to_nhwc(flo)))
