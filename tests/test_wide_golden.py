"""Reference-run fixtures at the hot-path widths (tests/golden/wide_*.npz, tests/wide_cases.py): the reference's own
CustomConvLayer / UpscalingLayer / ChengResBlock / SimplifiedAttention at 64 / 128 channels and its
InputLayer -> cat -> first analysis layer chain on 1 / 2 / 3 images
(src/layers/misc/custom_conv_layers.py:129-253, attention.py:22-97, src/layers/ae/ae_layers.py:27-35).

CPU half (this file, not gpu): the oracle against the reference's outputs.
GPU half (test_gpu_wide_* below, -m gpu): the aivc_amd.layers classes on cuda against the SAME outputs with no oracle
in the chain, asserting WHICH kernel instantiation ran (the ones the 1080p bench spends its time in: LDS-DMA K loop
on the 64x128 / 64x64 / 128x64 tiles with fused GDN, the fused 1x1 tail, aivc_conv_images, the thin MFMA kernel) and
re-running every conv case on the other tiles of the menu.

Tolerance 2e-5 (relative to max(1, |y|)): the reference's ATen / oneDNN kernels sum in another order than the
k-ordered fmaf chain; reductions here are up to 3200 terms long."""
import ast
import os

import numpy as np
import pytest
import torch

import wide_cases
from oracle import spec as ospec

NAMES = [c[0] for c in wide_cases.CASES]


def _build(name):
    """the aivc_amd module of a case with the seeded parameters loaded -> (module, input, fixture sha check)"""
    from aivc_amd.layers.misc import attention, custom_conv_layers as ccl
    _, build, kw, _, _, _, _ = wide_cases.CASE[name]
    if build == 'first_layer':
        m = ccl.CustomConvLayer(k_size=5, in_ft=3 * kw['n_img'], out_ft=64, non_linearity='gdn', conv_stride=2)
    else:
        m = (getattr(attention, build, None) or getattr(ccl, build))(**kw)
    x, sha = wide_cases.load_seeded(m.eval(), name)
    return m, x, sha


def _close(y, ref, tol=2e-5):
    err = np.abs(y - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol, 'max relative error %.3g' % err.max()


@pytest.mark.parametrize('name', NAMES)
def test_oracle_matches_reference_at_hot_path_widths(name, oracle, golden):
    g = golden('wide_' + name)
    m, x, sha = _build(name)
    assert sha == str(g['sha256']), 'the seeded draw changed: regenerate with tools/gen_golden.py'
    assert ast.literal_eval(str(g['cfg']))['seed'] == wide_cases.CASE[name][4]
    sp = ospec.export_spec(m)
    if isinstance(x, list):  # 8-bit planes -> the codec's stored image layout (3 channels padded to 4)
        h, w = x[0]['y'].shape[1:]
        cmap = oracle.image_cmap(len(x))
        y = oracle.run_layer(sp, np.ascontiguousarray(oracle.pack_images(x, h, w)[..., list(cmap)]), cmap=cmap)
        # and the twin of aivc_conv_images (no packed tensor) gives the very same bits
        be, ge = oracle.gdn_reparam(*(sp['gdn'][k] for k in ('beta', 'gamma', 'beta_bound', 'gamma_bound', 'pedestal')))
        wp = np.zeros((64, 5, 5, 4 * len(x)), np.float32)
        wp[..., list(cmap)] = oracle.pack_weight(sp['weight'])
        y2 = oracle.conv_images(x, h, w, wp, sp.get('bias'), gdn=(be, ge, False))
        np.testing.assert_array_equal(y, y2)
    else:
        y = oracle.run_layer(sp, np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1))))
    _close(np.transpose(y, (0, 3, 1, 2)), g['y'])


# ---- GPU half -------------------------------------------------------------------------------------------------------
def _run_gpu(m, x, cuda):
    """-> (NCHW numpy output, set of kernel variants the launches took)"""
    from aivc_amd import ops
    m = m.to(cuda)
    ops.PROFILE = []
    try:
        with torch.no_grad():
            if isinstance(x, list):
                h, w = x[0]['y'].shape[1:]
                parts = [{k: torch.from_numpy(p[k]).to(cuda) for k in 'yuv'} for p in x]
                y = ops.to_nchw_view(m.forward_nhwc(ops.ImageStack(parts, h, w, cuda)))
            else:
                y = m(torch.from_numpy(x).to(cuda))
        torch.cuda.synchronize()
        variants = {rec[0] for rec in ops.PROFILE}
    finally:
        ops.PROFILE = None
    return y.cpu().numpy(), variants


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_gpu_wide_layer_matches_reference_on_the_bench_kernels(name, cuda, golden):
    g = golden('wide_' + name)
    m, x, sha = _build(name)
    assert sha == str(g['sha256'])
    want = wide_cases.CASE[name][5]
    y, variants = _run_gpu(m, x, cuda)
    assert want <= variants, 'expected kernel instantiations %s, launches took %s' % (sorted(want), sorted(variants))
    assert 0 not in variants and 1 not in variants  # never the scalar / VALU fallbacks at these widths
    _close(y, g['y'])


@pytest.mark.gpu
@pytest.mark.parametrize('name,tile', [(c[0], t) for c in wide_cases.CASES for t in c[6]])
def test_gpu_wide_layer_matches_reference_on_every_tile(name, tile, cuda, golden):
    """the same case forced onto another tile of the MFMA menu (0 = 128x128, 1 = 64x64, 2 = 256x64, 5 = 64x128,
    6 = 128x64): every instantiation meets the reference directly, not only the one the tile rules pick at this size"""
    g = golden('wide_' + name)
    m, x, _ = _build(name)
    auto = next(iter(wide_cases.CASE[name][5]))
    os.environ['AIVC_FORCE_TILE'] = str(tile)
    try:
        y, variants = _run_gpu(m, x, cuda)
    finally:
        del os.environ['AIVC_FORCE_TILE']
    assert (auto // 10) * 10 + tile in variants, (sorted(variants), tile)
    _close(y, g['y'])
