"""The torch-CPU conv backend of the oracle (bench.py's cpu_baseline, oracle/torch_cpu.py): same layer specs, the
reference's torch ops (custom_conv_layers.py:129-253, misc_layers.py:113-154, attention.py:22-97) -- agrees with the
fixed-order fmaf oracle to fp32 rounding and is closed-loop consistent with itself."""
import numpy as np

from oracle import codec as ocodec
from oracle import spec as ospec
from oracle import torch_cpu


def _tiny():
    from aivc_amd import synth
    from aivc_amd.models import arch
    return synth.make_model(arch.TINY_WIDTHS, seed=5)


def test_transforms_agree_with_fmaf_oracle(oracle):
    spec = ospec.export_model(_tiny())
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (1, 40, 56, 6)).astype(np.float32)
    y = rng.standard_normal((1, 3, 4, 16)).astype(np.float32)
    for tr, inp, cmap in (('g_a', x, oracle.image_cmap(2)), ('g_s', y, None), ('h_a', y[..., :8], None),
                          ('h_s', rng.integers(-3, 4, (1, 1, 1, 4)).astype(np.float32), None)):
        want = oracle.run_layer(spec['cod'][tr], inp, cmap=cmap)
        got = torch_cpu.run_layer(spec['cod'][tr], inp)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * float(np.abs(want).max()))


def test_closed_loop_and_restore(oracle):
    from aivc_amd import synth
    spec = ospec.export_model(_tiny())
    frames = synth.synthetic_video(48, 32, 3, seed=3)
    keep = oracle.run_layer
    with torch_cpu.torch_convs(2):
        blob, recs = ocodec.encode_video(spec, frames, '1_GOP_2')
        dec = ocodec.decode_video(spec, blob)
    assert oracle.run_layer is keep
    assert all(np.array_equal(d[k], r[k]) for d, r in zip(dec, recs) for k in 'yuv')
