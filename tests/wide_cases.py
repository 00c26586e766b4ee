"""Layer cases at the HOT-PATH widths (64 / 128 channels, the widths the 1080p bench runs at) for the
reference-run fixtures tests/golden/wide_*.npz (tools/gen_golden.py) and their tests.

Parameters and inputs of these cases are NOT stored (a single SimplifiedAttention(128) carries 7 MB of
weights): generator and tests both draw them from numpy's frozen legacy generator (RandomState: bit-stable
across numpy versions by numpy's compatibility policy) with the seeds below, and every fixture carries the
sha256 of the exact tensors the reference ran on, so a drift of the draw is detected instead of showing up as
a numerics failure.  What IS stored is the output the reference's own module computed for them.

`build` names a class of layers.misc.* that exists under the same name in the reference and in aivc_amd;
`first_layer_*` is the chain InputLayer -> torch.cat -> CustomConvLayer(5, 3k -> 64, stride 2, gdn) the codec
runs on 1 / 2 / 3 images held as 8-bit 4:2:0 planes (src/layers/ae/ae_layers.py:17-35,
src/real_life/decode.py:631-636,709-714).

`variants`: the kernel instantiations (aivc_conv2d_variant codes; 191 = aivc_conv_images, 2 = thin MFMA
kernel) the product must take for the case with its automatic dispatch; `force_tiles`: other tiles of the
MFMA menu (AIVC_FORCE_TILE ids) the same case is additionally run on, each against the same reference output."""
import hashlib

import numpy as np

# name, build, kwargs, input shape (NCHW; for first_layer: (h, w)), seed, variants, force_tiles
CASES = [
    ('conv5_64_128_s2_gdn', 'CustomConvLayer', dict(k_size=5, in_ft=64, out_ft=128, non_linearity='gdn', conv_stride=2),
     (1, 64, 40, 54), 101, {155}, (0,)),
    ('conv5_128_64_s2_no', 'CustomConvLayer', dict(k_size=5, in_ft=128, out_ft=64, non_linearity='no', conv_stride=2),
     (1, 128, 23, 31), 102, {101}, (6, 2)),
    ('conv3_64_128_leaky', 'CustomConvLayer', dict(k_size=3, in_ft=64, out_ft=128, non_linearity='leaky_relu', conv_stride=1),
     (1, 64, 9, 13), 103, {101}, (0, 5)),  # (few tiles: the 64x64 tile; the bench's batches take 105 = tile 5, forced here)
    ('conv3_128_128_gdn', 'CustomConvLayer', dict(k_size=3, in_ft=128, out_ft=128, non_linearity='gdn', conv_stride=1),
     (1, 128, 11, 9), 104, {155}, (0,)),
    ('up5_128_64_igdn', 'UpscalingLayer', dict(k_size=5, in_ft=128, out_ft=64, non_linearity='gdn_inverse'),
     (1, 128, 11, 14), 105, {161}, (6, 2)),
    ('up5_128_128_igdn', 'UpscalingLayer', dict(k_size=5, in_ft=128, out_ft=128, non_linearity='gdn_inverse'),
     (1, 128, 9, 12), 106, {165}, (0,)),
    ('up5_32_128_leaky', 'UpscalingLayer', dict(k_size=5, in_ft=32, out_ft=128, non_linearity='leaky_relu'),
     (1, 32, 5, 8), 107, {111}, (0, 5)),
    ('up5_64_3_no', 'UpscalingLayer', dict(k_size=5, in_ft=64, out_ft=3, non_linearity='no'),
     (1, 64, 21, 37), 108, {2}, ()),
    ('up5_64_6_no', 'UpscalingLayer', dict(k_size=5, in_ft=64, out_ft=6, non_linearity='no'),
     (1, 64, 10, 35), 109, {2}, ()),
    ('cheng128_down', 'ChengResBlock', dict(nb_ft=128, mode='down'), (1, 128, 21, 30), 110, {101, 155}, ()),
    ('cheng128_up', 'ChengResBlock', dict(nb_ft=128, mode='up_tconv'), (1, 128, 10, 13), 111, {111, 155}, ()),
    ('attention128_light', 'SimplifiedAttention', dict(nb_ft=128, lightweight_resblock=True), (1, 128, 12, 17), 112,
     {190, 101}, ()),
    ('attention128_full', 'SimplifiedAttention', dict(nb_ft=128, lightweight_resblock=False), (1, 128, 9, 14), 113,
     {101}, ()),
    ('first_layer_1', 'first_layer', dict(n_img=1), (45, 67), 114, {191}, ()),
    ('first_layer_2', 'first_layer', dict(n_img=2), (46, 70), 115, {191}, ()),
    ('first_layer_3', 'first_layer', dict(n_img=3), (34, 52), 116, {151}, (6,)),  # (the bench's batches take 156 = tile 6)
]

CASE = {c[0]: c for c in CASES}


def seeded_arrays(state_dict_shapes, in_shape, seed, build):
    """(params {name: float32 array}, input) drawn in a fixed order from RandomState(seed).
    state_dict_shapes: ordered {name: (shape, initial value array)}; conv weights ~ N(0, 1/fan_in), biases
    N(0, 0.1), GDN beta / gamma = their initial value + U(0, .5) / U(0, .05) with a few entries pushed below the
    re-parameterisation bounds (src/layers/misc/misc_layers.py:131-149)."""
    rs = np.random.RandomState(seed)
    params = {}
    for name in sorted(state_dict_shapes):  # sorted: the draw does not depend on attribute definition order
        shape, init = state_dict_shapes[name]
        if name.endswith('beta'):
            a = init + rs.uniform(0., .5, shape)
            a[-1] = 0.
        elif name.endswith('gamma'):
            a = init + rs.uniform(0., .05, shape)
            a[0, -1] = 0.
        elif name.endswith('bias'):
            a = rs.standard_normal(shape) * .1
        else:
            fan = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            a = rs.standard_normal(shape) / np.sqrt(max(1, fan))
        params[name] = np.ascontiguousarray(a, np.float32)
    if build == 'first_layer':
        h, w = in_shape
        hc, wc = (h + 1) // 2, (w + 1) // 2
        n_img = state_dict_shapes['layers.1.weight'][0][1] // 3  # Conv2d weight [64, 3 * n_img, 5, 5]
        x = [{'y': rs.randint(0, 256, (1, h, w)).astype(np.uint8), 'u': rs.randint(0, 256, (1, hc, wc)).astype(np.uint8),
              'v': rs.randint(0, 256, (1, hc, wc)).astype(np.uint8)} for _ in range(n_img)]
    else:
        x = np.ascontiguousarray(rs.standard_normal(in_shape), np.float32)
    return params, x


def digest(params, x):
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(k.encode())
        h.update(params[k].tobytes())
    for part in (x if isinstance(x, list) else [x]):
        if isinstance(part, dict):
            for k in 'yuv':
                h.update(part[k].tobytes())
        else:
            h.update(part.tobytes())
    return h.hexdigest()


def shapes_of(module):
    """ordered {name: (shape, initial value)} of a torch module's state_dict (reference's or aivc_amd's: same names)"""
    return {k: (tuple(v.shape), v.detach().cpu().numpy().astype(np.float64)) for k, v in module.state_dict().items()}


def load_seeded(module, name):
    """Draw the case's parameters and input, load the parameters into `module` (strict).  -> (input, sha256)"""
    import torch
    _, build, _, in_shape, seed, _, _ = CASE[name]
    params, x = seeded_arrays(shapes_of(module), in_shape, seed, build)
    missing, unexpected = module.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    assert not missing and not unexpected
    return x, digest(params, x)
