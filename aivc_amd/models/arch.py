"""Network architecture of the build-authored ``models`` package.

The reference snapshot does not contain ``src/models/`` nor any weights (F2), so the
transform stacks below are this build's own arrangement of the reference's layer vocabulary
(CustomConvLayer / UpscalingLayer / ChengResBlock / SimplifiedAttention, 16x spatial reduction for
y, 64x for z, conditional "shortcut" transform g_a_ref concatenated in front of g_s).  The
architecture is *data*: every kernel launch is derived from the module tree, so a real pickled
model with other widths/depths runs through the same code.

Default widths ("ms_ssim-4" synthetic stand-in):  N2 = 64 features at 1/2 resolution, N = 128
deeper, C_y = 64 latent maps, C_shortcut = 64, C_z = 32, hyper width 128.
"""
from torch.nn import Sequential

from ..layers.misc.attention import SimplifiedAttention
from ..layers.misc.custom_conv_layers import ChengResBlock, CustomConvLayer, UpscalingLayer

DEFAULT_WIDTHS = {'n2': 64, 'n': 128, 'c_y': 64, 'c_short': 64, 'c_z': 32, 'n_h': 128}
TINY_WIDTHS = {'n2': 8, 'n': 16, 'c_y': 8, 'c_short': 8, 'c_z': 4, 'n_h': 8}  # unit tests / smoke


def analysis_transform(in_c, out_c, wd):
    """in_c @ HxW -> out_c @ H/16 x W/16"""
    return Sequential(
        CustomConvLayer(5, in_c, wd['n2'], non_linearity='gdn', conv_stride=2),
        CustomConvLayer(5, wd['n2'], wd['n'], non_linearity='gdn', conv_stride=2),
        ChengResBlock(wd['n'], mode='down'),
        SimplifiedAttention(wd['n'], lightweight_resblock=True),
        CustomConvLayer(5, wd['n'], out_c, non_linearity='no', conv_stride=2),
    )


def synthesis_transform(in_c, out_c, wd):
    """in_c @ H/16 -> out_c @ H"""
    return Sequential(
        SimplifiedAttention(in_c, lightweight_resblock=False),
        UpscalingLayer(5, in_c, wd['n'], non_linearity='gdn_inverse'),
        ChengResBlock(wd['n'], mode='up_tconv'),
        UpscalingLayer(5, wd['n'], wd['n2'], non_linearity='gdn_inverse'),
        UpscalingLayer(5, wd['n2'], out_c, non_linearity='no'),
    )


def hyper_analysis(c_y, c_z, wd):
    return Sequential(
        CustomConvLayer(3, c_y, wd['n_h'], non_linearity='leaky_relu'),
        CustomConvLayer(5, wd['n_h'], wd['n_h'], non_linearity='leaky_relu', conv_stride=2),
        CustomConvLayer(5, wd['n_h'], c_z, non_linearity='no', conv_stride=2),
    )


def hyper_synthesis(c_z, c_y, wd):
    return Sequential(
        UpscalingLayer(5, c_z, wd['n_h'], non_linearity='leaky_relu'),
        UpscalingLayer(5, wd['n_h'], wd['n_h'], non_linearity='leaky_relu'),
        CustomConvLayer(3, wd['n_h'], 2 * c_y, non_linearity='no'),
    )
