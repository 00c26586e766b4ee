"""HIP product path against the REFERENCE's own outputs, directly (no oracle in the chain): every layer
fixture of tests/golden (inputs, parameters and outputs of the reference's modules, tools/gen_golden.py) goes
through the aivc_amd.layers.* / func_util class of the same name on cuda, called through its module API
(forward on NCHW tensors), and is compared with the reference's result.

Tolerance 2e-5 (fp32): the reference runs ATen/oneDNN kernels whose summation order differs from the k-ordered
fmaf chain of the HIP kernels.  Pure data movement and integer results are exact."""
import ast

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(module, g, cuda):
    sd = {k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith('sd.')}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return module.eval().to(cuda)


def _cfg(g):
    return ast.literal_eval(str(g['cfg']))


def _check(module, g, cuda, rtol=2e-5, atol=2e-5):
    with torch.no_grad():
        y = module(torch.from_numpy(g['x']).to(cuda))
    assert y.is_cuda and tuple(y.shape) == g['y'].shape
    np.testing.assert_allclose(y.cpu().numpy(), g['y'], rtol=rtol, atol=atol)


@pytest.mark.parametrize('i', range(6))
def test_custom_conv_layer(i, cuda, golden):
    from aivc_amd.layers.misc.custom_conv_layers import CustomConvLayer
    g = golden('custom_conv_%d' % i)
    _check(_load(CustomConvLayer(**_cfg(g)), g, cuda), g, cuda)


@pytest.mark.parametrize('i', range(4))
def test_upscaling_layer(i, cuda, golden):
    from aivc_amd.layers.misc.custom_conv_layers import UpscalingLayer
    g = golden('upscaling_%d' % i)
    _check(_load(UpscalingLayer(**_cfg(g)), g, cuda), g, cuda)


@pytest.mark.parametrize('i', range(4))
def test_cheng_res_block(i, cuda, golden):
    """cheng_0 is mode 'plain' (src/layers/misc/custom_conv_layers.py:21-60), 1/2 'down', 3 'up_tconv'"""
    from aivc_amd.layers.misc.custom_conv_layers import ChengResBlock
    g = golden('cheng_%d' % i)
    assert i != 0 or _cfg(g)['mode'] == 'plain'
    _check(_load(ChengResBlock(**_cfg(g)), g, cuda), g, cuda)


def test_res_block(cuda, golden):
    from aivc_amd.layers.misc.custom_conv_layers import ResBlock
    g = golden('resblock_0')
    _check(_load(ResBlock(**_cfg(g)), g, cuda), g, cuda)


@pytest.mark.parametrize('i', range(2))
def test_simplified_attention(i, cuda, golden):
    from aivc_amd.layers.misc.attention import SimplifiedAttention
    g = golden('attention_%d' % i)
    _check(_load(SimplifiedAttention(**_cfg(g)), g, cuda), g, cuda)


@pytest.mark.parametrize('i', range(2))
def test_gdn(i, cuda, golden):
    from aivc_amd.layers.misc.misc_layers import GDN
    g = golden('gdn_%d' % i)
    _check(_load(GDN(**_cfg(g)), g, cuda), g, cuda)


@pytest.mark.parametrize('i', range(3))
def test_input_output_layers(i, cuda, golden):
    from aivc_amd import ops
    from aivc_amd.layers.ae.ae_layers import InputLayer, OutputLayer
    g = golden('inout_layer_%d' % i)
    d = {k: torch.from_numpy(g[k]).to(cuda) for k in 'yuv'}
    with torch.no_grad():
        x444 = InputLayer()(d)
        np.testing.assert_array_equal(x444.cpu().numpy(), g['x444'])
        h, w = g['y'].shape[2:]
        z = torch.from_numpy(g['z']).to(cuda)
        o = OutputLayer()(z[:, :, :h, :w])
        assert o['u'].shape[2:] == (h // 2, w // 2)
        # OutputLayer + replicate pad + crop + 8-bit cast as Decoder.decode chains them (decode.py:553-577)
        (fy, fu, fv), (y8, u8, v8) = ops.frame_to_yuv420(ops.to_nhwc(z), h, w)
    for got, b, ref in ((fy, y8, g['oy']), (fu, u8, g['ou']), (fv, v8, g['ov'])):
        np.testing.assert_array_equal(got.cpu().numpy()[:, None], ref)
        np.testing.assert_array_equal(b.cpu().numpy(), np.rint(ref[:, 0] * 255).astype(np.uint8))
    np.testing.assert_allclose(o['u'].cpu().numpy(), 0.25 * (g['z'][:, 1:2, 0:h - h % 2:2, 0:w - w % 2:2]
                                                              + g['z'][:, 1:2, 1:h:2, 0:w - w % 2:2]
                                                              + g['z'][:, 1:2, 0:h - h % 2:2, 1:w:2]
                                                              + g['z'][:, 1:2, 1:h:2, 1:w:2]), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('i', range(3))
def test_warp(i, cuda, golden):
    from aivc_amd.func_util.optical_flow import warp
    g = golden('warp_%d' % i)
    with torch.no_grad():
        y = warp(torch.from_numpy(g['x']).to(cuda), torch.from_numpy(g['flow']).to(cuda))
    np.testing.assert_allclose(y.cpu().numpy(), g['y'], rtol=1e-5, atol=2e-6)


def test_pdf_param_parameterizer(cuda, golden):
    from aivc_amd.layers.misc.misc_layers import PdfParamParameterizer
    g = golden('pdf_param_0')
    with torch.no_grad():
        r = PdfParamParameterizer('laplace', 6)(torch.from_numpy(g['x']).to(cuda))
    np.testing.assert_array_equal(r[0]['mu'].cpu().numpy(), g['mu'])
    np.testing.assert_allclose(r[0]['sigma'].cpu().numpy(), g['sigma'], rtol=2e-7, atol=0)


def test_gain_matrix(cuda, golden):
    from aivc_amd.layers.multi_rate.gain_matrix import GainMatrix
    g = golden('gain_matrix_0')
    gm = _load(GainMatrix({'N': 3, 'nb_ft': 6, 'initialize_to_one': False}), g, cuda)
    x = torch.from_numpy(g['x']).to(cuda)
    for idx in (0, 1, 2, 0.5, 1.25):
        for mode in ('enc', 'dec'):
            with torch.no_grad():
                y = gm({'x': x, 'idx_rate': idx, 'mode': mode})['output'].cpu().numpy()
            ref = g['y_%s_%s' % (str(idx).replace('.', 'p'), mode)]
            if float(idx) == int(idx):
                np.testing.assert_array_equal(y, ref)
            else:
                np.testing.assert_allclose(y, ref, rtol=1e-6, atol=1e-7)


def test_quantizer(cuda, golden):
    from aivc_amd.layers.misc.misc_layers import Quantizer
    g = golden('quantizer_0')
    with torch.no_grad():
        y = Quantizer().eval()(torch.from_numpy(g['x']).to(cuda)).cpu().numpy().reshape(-1)
    ref = g['y'].reshape(-1)
    ok = ref <= 256  # values are clamped to the coder's alphabet [-256, 256] (symbols 0 .. 512)
    np.testing.assert_array_equal(y[ok], ref[ok])


@pytest.mark.parametrize('i', range(2))
def test_balle_cdf_table(i, cuda, golden):
    from aivc_amd.layers.entropy_coding.pdf_estimator import BallePdfEstim
    from aivc_amd.real_life.bitstream import ArithmeticCoder
    g = golden('balle_cdf_%d' % i)
    cz = g['cdf'].shape[0]
    pe = _load(BallePdfEstim(cz, 'balle', verbose=False), g, cuda)
    ac = ArithmeticCoder({'balle_pdf_estim_z': pe, 'device': cuda})
    with torch.no_grad():
        cdf = ac.pre_computed_z_cdf
        table = ac.z_table(cuda)
    assert tuple(cdf.shape) == (1, cz, 1, 1, 514)  # the reference's attribute layout (bitstream.py:82-125)
    np.testing.assert_allclose(cdf.reshape(cz, 514).cpu().numpy(), g['cdf'], rtol=0, atol=3e-7)
    ref16 = ((np.rint(g['cdf'] * np.float32(65023)).astype(np.int64) + np.arange(514)) & 0xFFFF)
    diff = np.abs(table.cpu().numpy().view(np.uint16)[:, :514].astype(np.int64) - ref16)
    assert diff.max() <= 1 and (diff != 0).mean() < 0.01


def test_rate_estimation_modules(cuda, golden, oracle):
    """EntropyCoder / ParametricPdf / BallePdfEstim forward (logging only) on the HIP kernels of csrc/rate.hip: against
    the reference's outputs (rate_est_0) and bit for bit against the oracle's twins, sums included"""
    from aivc_amd import ops
    from aivc_amd.func_util.math_func import PROBA_MAX, PROBA_MIN
    from aivc_amd.layers.entropy_coding.entropy_coder import EntropyCoder
    from aivc_amd.layers.entropy_coding.pdf_estimator import BallePdfEstim, ParametricPdf
    g = golden('rate_est_0')
    t = lambda k: torch.from_numpy(np.asarray(g[k])).to(cuda)
    y, mu, sigma, xz = t('y'), t('mu'), t('sigma'), t('xz')
    pp = ParametricPdf('laplace')
    with torch.no_grad():
        p_mu = pp(y, [{'mu': mu, 'sigma': sigma}])
        p_zero = pp(y, [{'mu': mu, 'sigma': sigma}], zero_mu=True)
        pe = _load(BallePdfEstim(xz.shape[1], 'balle', verbose=False), g, cuda)
        p_z = pe(xz)
        ec = EntropyCoder()
        rate_y, rate_z = ec(p_zero, y), ec(p_z, xz)
    np.testing.assert_allclose(p_mu.cpu().numpy(), g['p_mu'], rtol=0, atol=3e-7)
    np.testing.assert_allclose(p_zero.cpu().numpy(), g['p_zero'], rtol=0, atol=3e-7)
    np.testing.assert_allclose(p_z.cpu().numpy(), g['p_z'], rtol=0, atol=6e-7)
    big = g['p_zero'] > 2.0 ** -10
    np.testing.assert_allclose(rate_y.cpu().numpy()[big], g['rate_y'][big], rtol=0, atol=2e-3)
    big = g['p_z'] > 2.0 ** -10
    np.testing.assert_allclose(rate_z.cpu().numpy()[big], g['rate_z'][big], rtol=0, atol=2e-3)
    # HIP == oracle, bit for bit
    assert np.array_equal(p_mu.cpu().numpy(), oracle.laplace_prob(g['y'], g['mu'], g['sigma']))
    assert np.array_equal(p_zero.cpu().numpy(), oracle.laplace_prob(g['y'], None, g['sigma']))
    _, cdf = pe.cdf_table(cuda, want_float=True)
    assert np.array_equal(p_z.cpu().numpy(), oracle.table_prob(g['xz'], cdf.cpu().numpy()))
    r_h, s_h = ops.rate_bits(p_zero, PROBA_MIN, PROBA_MAX)
    r_o, s_o = oracle.rate_bits(p_zero.cpu().numpy(), PROBA_MIN, PROBA_MAX)
    assert np.array_equal(r_h.cpu().numpy(), r_o) and float(s_h.cpu()) == s_o
    # values that are not codable symbols
    bad = torch.tensor([[[[0.5, 300.0, -257.0, 2.0]]]], device=cuda).expand(1, xz.shape[1], 1, 4).contiguous()
    out = pe(bad).cpu().numpy()
    assert np.isnan(out[..., :3]).all() and np.isfinite(out[..., 3]).all()


def test_bounds_rate_hip_equals_oracle(cuda, oracle):
    from aivc_amd import ops
    rng = np.random.default_rng(11)
    for n in (0, 1, 1000, 16384, 16385, 200001):
        lo = rng.integers(0, 0xFFFF, n)
        hi = lo + 1 + (rng.integers(0, 0x10000, n) % (0x10000 - lo))
        b = (lo | ((hi & 0xFFFF) << 16)).astype(np.uint32)
        got = ops.bounds_rate(torch.from_numpy(b.view(np.int32)).to(cuda))
        assert float(got.cpu()) == oracle.bounds_rate(b), n
