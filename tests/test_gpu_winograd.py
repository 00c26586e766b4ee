"""Version 2 of the fp32 arithmetic contract (AIVC_PREC_FP32_WINO, include/aivc_hip.h): the stride-1 3x3 layers with
c_in % 32 == 0 and c_out % 128 == 0 on Winograd F(2x2, 3x3) chains (csrc/conv_wino.hip).  Checked here:
  * HIP == CPU oracle BIT FOR BIT (the oracle walks the same chain: oracle/aivc_oracle.c) on odd sizes, batches, every
    epilogue combination, both tile widths; the weight transform likewise;
  * the error against an fp64 evaluation next to version 1's (both are fp32 summation noise);
  * layers the version does not cover are version 1's bits unchanged."""
import numpy as np
import pytest
import torch

from aivc_amd import abi

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


CASES = [  # n, h, w, c_in, c_out, act1, act2, bias, mul, res
    (1, 8, 8, 32, 128, 0, 0, True, False, False),
    (2, 7, 9, 32, 128, 1, 0, True, False, False),      # odd sizes: half-filled last tile row / column
    (3, 13, 21, 64, 128, 0, 2, True, False, True),     # residual + relu
    (1, 17, 30, 128, 128, 1, 0, True, False, True),    # leaky then residual (ChengResBlock)
    (5, 5, 3, 128, 128, 0, 1, False, True, True),      # no bias, gate multiplicand, tiny images: a tile spans images
    (1, 1, 1, 32, 128, 0, 0, True, False, False),      # a single pixel
    (2, 34, 60, 128, 128, 0, 0, True, False, False),
    (1, 9, 40, 64, 256, 2, 0, True, False, False),     # c_out 256: four 64-channel blocks
    (1, 68, 120, 128, 128, 0, 0, True, False, True),   # the 1/16-resolution shape of 1080p
    (2, 33, 50, 64, 128, 1, 0, True, False, True),     # interior blocks (lean epilogue) + right-edge / bottom-edge blocks
    (1, 47, 64, 32, 128, 0, 1, True, False, False),    # leaky after the (absent) residual
    (1, 32, 48, 32, 256, 2, 0, False, False, False),   # relu, no bias
]


def _inputs(case, seed):
    n, h, w, ci, co, a1, a2, has_b, has_m, has_r = case
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, h, w, ci)).astype(np.float32)
    wt = (rng.standard_normal((co, 3, 3, ci)) / np.sqrt(9 * ci)).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32) if has_b else None
    m = rng.standard_normal((n, h, w, co)).astype(np.float32) if has_m else None
    r = rng.standard_normal((n, h, w, co)).astype(np.float32) if has_r else None
    return x, wt, b, m, r


@pytest.fixture()
def fp32w(oracle):
    from aivc_amd import ops
    prev_h, prev_o = ops.set_precision('fp32w'), oracle.set_precision('fp32w')
    ops.WINO_ANY_SIZE = oracle.WINO_ANY_SIZE = True  # the kernel on shapes the oracle checks in seconds
    yield
    ops.WINO_ANY_SIZE = oracle.WINO_ANY_SIZE = False
    ops.set_precision(prev_h)
    oracle.set_precision(prev_o)


@pytest.mark.parametrize('idx', range(len(CASES)))
def test_hip_equals_oracle_bit_for_bit(idx, cuda, oracle, fp32w):
    from aivc_amd import ops
    case = CASES[idx]
    n, h, w, ci, co, a1, a2, _, _, _ = case
    x, wt, b, m, r = _inputs(case, 100 + idx)
    want = oracle.conv2d(x, wt, b, stride=1, pad=1, act1=a1, act2=a2, mul=m, res=r)
    dv = lambda a: None if a is None else T(a, cuda)
    ops.PROFILE = []
    try:
        got = ops.conv2d(dv(x), dv(wt), dv(b), stride=1, pad=1, act1=a1, act2=a2, mul=dv(m), res=dv(r))
        torch.cuda.synchronize()
        variants = [pr[0] for pr in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert variants == [301], variants
    assert np.array_equal(got.cpu().numpy(), want), (case, float(np.abs(got.cpu().numpy() - want).max()))


def test_weight_transform_equals_oracle(cuda, oracle):
    from aivc_amd import ops
    rng = np.random.default_rng(9)
    w = (rng.standard_normal((64, 3, 3, 96)) * 3).astype(np.float32)  # (the transform itself takes any c_out % 64 == 0)
    u = ops.winograd_weights(T(w, cuda)).cpu().numpy()
    assert np.array_equal(u, oracle.winograd_weights(w))
    # G g G^T in fp64
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    ref = np.einsum('ik,oklc,jl->oijc', G, w.astype(np.float64), G).reshape(64, 16, 96)
    # the staging order of include/aivc_hip.h (AIVC_WINO_U_INDEX): [co / 64][ci / 8][p][(ci % 8) / 4][co % 64][ci % 4]
    img = u.reshape(1, 12, 16, 2, 64, 4)
    back = np.transpose(img, (0, 4, 2, 1, 3, 5)).reshape(64, 16, 96)
    assert np.abs(back - ref).max() <= np.abs(ref).max() * 2.0 ** -23


def test_error_against_fp64_next_to_version_1(cuda, oracle):
    """the two versions of the contract against an fp64 evaluation of a 3x3 128 -> 128 layer: both are summation noise of an
    fp32 accumulator; version 2's per-position chains are 9 times shorter"""
    from aivc_amd import ops
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1, 40, 52, 128)).astype(np.float32) * 3
    wt = (rng.standard_normal((128, 3, 3, 128)) / np.sqrt(9 * 128)).astype(np.float32)
    b = (rng.standard_normal(128) * 0.1).astype(np.float32)
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, (1, 1, 1, 1), mode='replicate'),
                                     torch.from_numpy(wt).double().permute(0, 3, 1, 2), torch.from_numpy(b).double())
    ref = ref.permute(0, 2, 3, 1).numpy()
    errs = {}
    for mode in ('fp32', 'fp32w'):
        prev = ops.set_precision(mode)
        ops.WINO_ANY_SIZE = True
        try:
            y = ops.conv2d(T(x, cuda), T(wt, cuda), T(b, cuda), stride=1, pad=1).cpu().numpy()
        finally:
            ops.WINO_ANY_SIZE = False
            ops.set_precision(prev)
        e = np.abs(y - ref) / np.maximum(1.0, np.abs(ref))
        errs[mode] = (float(e.max()), float(np.sqrt((e ** 2).mean())))
    print('\nmax / rms error vs fp64: version 1 %.2e / %.2e, version 2 (Winograd) %.2e / %.2e' % (errs['fp32'] + errs['fp32w']))
    assert errs['fp32w'][0] <= max(2e-5, 4 * errs['fp32'][0]) and errs['fp32w'][1] <= 3 * errs['fp32'][1]


def test_uncovered_layers_keep_version_1_bits(cuda, fp32w):
    """stride 2, 5x5, 1x1, c_in not a multiple of 32, transposed: version 2 is version 1 there"""
    from aivc_amd import ops
    rng = np.random.default_rng(12)
    for mode, k, s, pad, ci, co in ((abi.MODE_CONV, 3, 2, 1, 64, 64), (abi.MODE_CONV, 5, 1, 2, 32, 64), (abi.MODE_CONV, 1, 1, 0, 64, 64),
                                    (abi.MODE_CONV, 3, 1, 1, 16, 128), (abi.MODE_TCONV, 3, 2, 0, 64, 128), (abi.MODE_CONV, 3, 1, 1, 64, 64)):
        x = T(rng.standard_normal((1, 10, 12, ci)).astype(np.float32), cuda)
        w = T((rng.standard_normal((co, k, k, ci)) / np.sqrt(k * k * ci)).astype(np.float32), cuda)
        y2 = ops.conv2d(x, w, None, mode=mode, stride=s, pad=pad)
        prev = ops.set_precision('fp32')
        try:
            y1 = ops.conv2d(x, w, None, mode=mode, stride=s, pad=pad)
        finally:
            ops.set_precision(prev)
        assert torch.equal(y1, y2), (mode, k, s, ci, co)


def test_fused_gdn_request_is_two_launches_with_the_oracle_bits(cuda, oracle, fp32w):
    """a covered layer with a fused (I)GDN: the Winograd launch then the GDN-mode launch (aivc_conv2d_variant says the
    fusion is not available in this version) == the oracle's fused evaluation"""
    from aivc_amd import ops
    rng = np.random.default_rng(21)
    c = 128
    x = rng.standard_normal((2, 11, 9, c)).astype(np.float32)
    wt = (rng.standard_normal((c, 3, 3, c)) / np.sqrt(9 * c)).astype(np.float32)
    b = (rng.standard_normal(c) * 0.1).astype(np.float32)
    beta = (1.0 + rng.uniform(0, .5, c)).astype(np.float32)
    gamma = (0.1 * np.eye(c) + rng.uniform(0, .05, (c, c))).astype(np.float32)
    res = rng.standard_normal((2, 11, 9, c)).astype(np.float32)
    for inverse in (False, True):
        want = oracle.conv2d(x, wt, b, stride=1, pad=1, gdn=(beta, gamma, inverse), res=res)
        got = ops.conv2d(T(x, cuda), T(wt, cuda), T(b, cuda), stride=1, pad=1, gdn=(T(beta, cuda), T(gamma, cuda), inverse), res=T(res, cuda))
        assert np.array_equal(got.cpu().numpy(), want), inverse


def test_size_rule_of_the_version(cuda, oracle):
    """aivc_winograd_covers: below AIVC_WINO_MIN_PIXELS input pixels a layer is version 1 in version 2 as well (the same rule
    on both sides: HIP == oracle); at the threshold it is the Winograd launch"""
    from aivc_amd import ops
    rng = np.random.default_rng(31)
    wt = (rng.standard_normal((128, 3, 3, 32)) / 17).astype(np.float32)
    prev_h, prev_o = ops.set_precision('fp32w'), oracle.set_precision('fp32w')
    try:
        for h, w, want in ((128, 128, 301), (68, 120, 301), (64, 124, 101)):
            x = rng.standard_normal((1, h, w, 32)).astype(np.float32)
            ops.PROFILE = []
            got = ops.conv2d(T(x, cuda), T(wt, cuda), None, stride=1, pad=1)
            torch.cuda.synchronize()
            variants = [pr[0] for pr in ops.PROFILE]
            ops.PROFILE = None
            assert (variants == [301]) == (want == 301), (h, w, variants)
            assert np.array_equal(got.cpu().numpy(), oracle.conv2d(x, wt, None, stride=1, pad=1)), (h, w)
    finally:
        ops.PROFILE = None
        ops.set_precision(prev_h)
        oracle.set_precision(prev_o)


def test_codec_in_version_2_equals_the_oracle_bytes_and_frames(cuda, oracle):
    """The whole codec (default widths: the layers version 2 covers run at 1/4 resolution, 128 x 128 = AIVC_WINO_MIN_PIXELS
    pixels for a 512 x 512 frame) in version 2 of the contract: an I + P + B triple coded by the HIP path and by the CPU oracle,
    both in 'fp32w' -- container bytes equal, decoded frames equal to the oracle's reconstruction, Winograd launches actually
    taken; and the bytes differ from version 1's (it is another contract: both ends must run the same one)."""
    from aivc_amd import ops, synth
    from aivc_amd.models import arch
    from oracle import codec as ocodec
    from oracle import spec as ospec
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=cuda)
    synth.calibrate_operating_point(model, cuda)
    frames = synth.synthetic_video(512, 512, 3, seed=9)
    fc = model.frame_codec()
    prev_h = ops.set_precision('fp32')
    try:
        with torch.no_grad():
            blob1 = fc.assemble_video(fc.encode_video(synth.to_device_frames(frames, cuda), '1_GOP_2'))  # version 1's bytes
    finally:
        ops.set_precision(prev_h)
    prev_h, prev_o = ops.set_precision('fp32w'), oracle.set_precision('fp32w')
    ops.PROFILE = []
    try:
        with torch.no_grad():
            enc = fc.encode_video(synth.to_device_frames(frames, cuda), '1_GOP_2')
            blob = fc.assemble_video(enc)
            dec, _, _, _ = fc.decode_video(blob, cuda)
        torch.cuda.synchronize()
        n_wino = sum(1 for pr in ops.PROFILE if pr[0] in (301, 302))
        ops.PROFILE = None
        assert fc.stream_errors() == []
        spec = ospec.export_model(model)
        ref_blob, ref_rec = ocodec.encode_video(spec, frames, '1_GOP_2')
    finally:
        ops.PROFILE = None
        ops.set_precision(prev_h)
        oracle.set_precision(prev_o)
    assert n_wino > 0, 'no layer of the codec took the Winograd launch'
    assert blob == ref_blob
    for d, r in zip(dec, ref_rec):
        for k in 'yuv':
            np.testing.assert_array_equal(d[k][0].cpu().numpy(), r[k])
    assert blob != blob1


def test_wide_reference_fixtures_in_version_2(cuda, golden):
    """the reference's own outputs at the hot-path widths (tests/golden/wide_*.npz: CustomConvLayer 3x3 128 -> 128 + GDN, the
    ChengResBlocks, the attention modules ...) with the covered layers on Winograd chains (size rule lifted: the fixtures are
    small): version 2 meets the bound the version 1 kernels are held to (2e-5 relative to max(1, |y|)); printed side by side"""
    from test_wide_golden import NAMES, _build, _run_gpu
    from aivc_amd import ops
    worst, took = {}, 0
    for name in NAMES:
        g = golden('wide_' + name)
        m, x, _ = _build(name)
        errs = []
        for mode in ('fp32', 'fp32w'):
            prev = ops.set_precision(mode)
            ops.WINO_ANY_SIZE = mode == 'fp32w'
            try:
                y, variants = _run_gpu(m, x, cuda)
            finally:
                ops.WINO_ANY_SIZE = False
                ops.set_precision(prev)
            errs.append(float((np.abs(y - g['y']) / np.maximum(1.0, np.abs(g['y']))).max()))
            if mode == 'fp32w':
                took += 1 if 301 in variants else 0
        worst[name] = errs
    print('\nmax relative error vs the reference outputs, version 1 | version 2 (Winograd where covered):')
    for k, (e0, e1) in worst.items():
        print('  %-22s %.2e | %.2e' % (k, e0, e1))
    assert took >= 3, 'only %d of the wide fixtures took a Winograd launch' % took
    assert max(e[1] for e in worst.values()) <= 2e-5, worst


# ---- the 5x5 stride-2 layers in polyphase form (ABI 17: four stride-1 3x3 convolutions of the input's phases, 49 multiplications per
# 2x2 outputs instead of 100; include/aivc_hip.h, aivc_winograd_covers) -------------------------------------------------------------
POLY_CASES = [  # n, h, w, c_in, c_out, act1, act2, bias, res
    (1, 8, 8, 32, 128, 0, 0, True, False),
    (2, 15, 17, 64, 128, 1, 0, True, False),     # odd input sizes: the last phase row / column clamps into the other phase's samples
    (1, 33, 47, 64, 128, 0, 2, True, True),
    (1, 64, 96, 64, 128, 0, 0, True, False),     # 32 x 48 outputs: 2 x 3 blocks
    (3, 9, 7, 32, 256, 2, 0, False, False),      # four 64-channel blocks, tiny images
    (1, 136, 240, 64, 128, 0, 0, True, True),    # 68 x 120 outputs: right-edge column, bottom block row mostly outside
    (1, 7, 5, 128, 128, 0, 1, True, True),
    (1, 1, 1, 32, 128, 0, 0, True, False),       # a single pixel: every tap clamps onto it
    (2, 34, 62, 64, 128, 1, 0, True, True),
]


@pytest.mark.parametrize('idx', range(len(POLY_CASES)))
def test_polyphase_5x5_stride_2_hip_equals_oracle_bit_for_bit(idx, cuda, oracle, fp32w):
    from aivc_amd import ops
    n, h, w, ci, co, a1, a2, has_b, has_r = POLY_CASES[idx]
    rng = np.random.default_rng(500 + idx)
    x = rng.standard_normal((n, h, w, ci)).astype(np.float32)
    wt = (rng.standard_normal((co, 5, 5, ci)) / np.sqrt(25 * ci)).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32) if has_b else None
    ho, wo = abi.conv_out_size(abi.MODE_CONV, h, w, 5, 2, 2)
    r = rng.standard_normal((n, ho, wo, co)).astype(np.float32) if has_r else None
    want = oracle.conv2d(x, wt, b, stride=2, pad=2, act1=a1, act2=a2, res=r)
    dv = lambda a: None if a is None else T(a, cuda)
    ops.PROFILE = []
    try:
        got = ops.conv2d(dv(x), dv(wt), dv(b), stride=2, pad=2, act1=a1, act2=a2, res=dv(r))
        torch.cuda.synchronize()
        variants = [pr[0] for pr in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert variants == [302], variants
    g = got.cpu().numpy()
    assert np.array_equal(g, want), (POLY_CASES[idx], float(np.abs(g - want).max()))
    # and within summation noise of version 1 (the tap chain) on the same inputs
    prev = ops.set_precision('fp32')
    try:
        v1 = ops.conv2d(dv(x), dv(wt), dv(b), stride=2, pad=2, act1=a1, act2=a2, res=dv(r)).cpu().numpy()
    finally:
        ops.set_precision(prev)
    assert np.abs(g - v1).max() <= 2e-5 * max(1.0, float(np.abs(v1).max()))


def test_polyphase_weight_transform_equals_oracle(cuda, oracle):
    from aivc_amd import ops
    rng = np.random.default_rng(19)
    w = (rng.standard_normal((64, 5, 5, 32)) * 3).astype(np.float32)
    u = ops.winograd_weights(T(w, cuda)).cpu().numpy()
    assert u.size == 64 * 16 * 4 * 32
    assert np.array_equal(u, oracle.winograd_weights(w))
    # positions that are zero by construction: i == 3 for py = 1, j == 3 for px = 1 (virtual channel = phase * c_in + ci)
    img = u.reshape(1, 16, 16, 2, 64, 4)  # [co / 64][cv / 8][p][(cv % 8) / 4][co % 64][cv % 4]
    for phase in range(4):
        py, px = phase >> 1, phase & 1
        chunks = img[:, 4 * phase:4 * phase + 4]
        for pos in range(16):
            i, j = pos >> 2, pos & 3
            zero = (py == 1 and i == 3) or (px == 1 and j == 3)
            assert (np.abs(chunks[:, :, pos]).max() == 0.0) == zero, (phase, pos)


def test_polyphase_error_against_fp64_next_to_version_1(cuda):
    from aivc_amd import ops
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 80, 104, 64)).astype(np.float32) * 3
    wt = (rng.standard_normal((128, 5, 5, 64)) / np.sqrt(25 * 64)).astype(np.float32)
    b = (rng.standard_normal(128) * 0.1).astype(np.float32)
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, (2, 2, 2, 2), mode='replicate'),
                                     torch.from_numpy(wt).double().permute(0, 3, 1, 2), torch.from_numpy(b).double(), stride=2)
    ref = ref.permute(0, 2, 3, 1).numpy()
    errs = {}
    for mode in ('fp32', 'fp32w'):
        prev = ops.set_precision(mode)
        ops.WINO_ANY_SIZE = True
        try:
            y = ops.conv2d(T(x, cuda), T(wt, cuda), T(b, cuda), stride=2, pad=2).cpu().numpy()
        finally:
            ops.WINO_ANY_SIZE = False
            ops.set_precision(prev)
        e = np.abs(y - ref) / np.maximum(1.0, np.abs(ref))
        errs[mode] = (float(e.max()), float(np.sqrt((e ** 2).mean())))
    print('\n5x5 stride 2, max / rms error vs fp64: version 1 %.2e / %.2e, polyphase Winograd %.2e / %.2e' % (errs['fp32'] + errs['fp32w']))
    assert errs['fp32w'][0] <= max(2e-5, 4 * errs['fp32'][0]) and errs['fp32w'][1] <= 3 * errs['fp32'][1]


def test_polyphase_fused_gdn_request_is_two_launches_with_the_oracle_bits(cuda, oracle, fp32w):
    from aivc_amd import ops
    rng = np.random.default_rng(23)
    x = rng.standard_normal((2, 21, 19, 64)).astype(np.float32)
    wt = (rng.standard_normal((128, 5, 5, 64)) / np.sqrt(25 * 64)).astype(np.float32)
    b = (rng.standard_normal(128) * 0.1).astype(np.float32)
    beta = (1.0 + rng.uniform(0, .5, 128)).astype(np.float32)
    gamma = (0.1 * np.eye(128) + rng.uniform(0, .05, (128, 128))).astype(np.float32)
    want = oracle.conv2d(x, wt, b, stride=2, pad=2, gdn=(beta, gamma, False))
    got = ops.conv2d(T(x, cuda), T(wt, cuda), T(b, cuda), stride=2, pad=2, gdn=(T(beta, cuda), T(gamma, cuda), False))
    assert np.array_equal(got.cpu().numpy(), want)


# ---- the transposed 5x5 stride-2 layers class by class (ABI 17: each output parity class a stride-1 3x3 correlation of the
# zero-extended input, 49 multiplications per 2x2 grid pixels instead of 100) ---------------------------------------------------------
TC_CASES = [  # n, h, w, c_in, c_out, act1, act2, bias, res
    (1, 8, 8, 32, 64, 0, 0, True, False),
    (2, 7, 9, 64, 64, 1, 0, True, False),        # odd sizes: half-filled last tile row / column, zero extension on every side
    (1, 17, 30, 128, 64, 0, 0, True, True),
    (1, 33, 50, 128, 128, 0, 2, True, True),     # c_out 128: two channel blocks per class; interior + edge blocks
    (3, 5, 3, 32, 128, 0, 0, False, False),      # tiny images: a block spans nothing but border
    (1, 68, 120, 128, 64, 0, 0, True, False),    # the 1/16-resolution shape of 1080p: right-edge column
    (1, 1, 1, 32, 64, 0, 0, True, False),
    (2, 34, 60, 64, 128, 2, 0, True, True),
]


@pytest.mark.parametrize('idx', range(len(TC_CASES)))
def test_transposed_5x5_stride_2_hip_equals_oracle_bit_for_bit(idx, cuda, oracle, fp32w):
    from aivc_amd import ops
    n, h, w, ci, co, a1, a2, has_b, has_r = TC_CASES[idx]
    rng = np.random.default_rng(700 + idx)
    x = rng.standard_normal((n, h, w, ci)).astype(np.float32)
    wt = (rng.standard_normal((co, 5, 5, ci)) / np.sqrt(6.25 * ci)).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32) if has_b else None
    r = rng.standard_normal((n, 2 * h, 2 * w, co)).astype(np.float32) if has_r else None
    want = oracle.conv2d(x, wt, b, mode=abi.MODE_TCONV, stride=2, act1=a1, act2=a2, res=r)
    dv = lambda a: None if a is None else T(a, cuda)
    ops.PROFILE = []
    try:
        got = ops.conv2d(dv(x), dv(wt), dv(b), mode=abi.MODE_TCONV, stride=2, act1=a1, act2=a2, res=dv(r))
        torch.cuda.synchronize()
        variants = [pr[0] for pr in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert variants == [303], variants
    g = got.cpu().numpy()
    assert np.array_equal(g, want), (TC_CASES[idx], float(np.abs(g - want).max()))
    prev = ops.set_precision('fp32')  # within summation noise of version 1 (the tap chains of the classes)
    try:
        v1 = ops.conv2d(dv(x), dv(wt), dv(b), mode=abi.MODE_TCONV, stride=2, act1=a1, act2=a2, res=dv(r)).cpu().numpy()
    finally:
        ops.set_precision(prev)
    assert np.abs(g - v1).max() <= 2e-5 * max(1.0, float(np.abs(v1).max()))


def test_transposed_weight_transform_equals_oracle(cuda, oracle):
    from aivc_amd import ops
    rng = np.random.default_rng(29)
    w = (rng.standard_normal((64, 5, 5, 32)) * 3).astype(np.float32)
    u = ops.winograd_weights(T(w, cuda), transposed=True).cpu().numpy()
    assert u.size == 4 * 64 * 16 * 32
    assert np.array_equal(u, oracle.winograd_weights(w, transposed=True))
    img = u.reshape(4, 4, 16, 2, 64, 4)  # [class (co_virtual / 64)][ci / 8][p][(ci % 8) / 4][co % 64][ci % 4]
    for cls in range(4):
        pyc, pxc = cls >> 1, cls & 1
        for pos in range(16):
            i, j = pos >> 2, pos & 3
            zero = (pyc == 1 and i == 0) or (pxc == 1 and j == 0)
            assert (np.abs(img[cls, :, pos]).max() == 0.0) == zero, (cls, pos)


def test_transposed_fused_igdn_request_is_two_launches_with_the_oracle_bits(cuda, oracle, fp32w):
    from aivc_amd import ops
    rng = np.random.default_rng(31)
    x = rng.standard_normal((2, 11, 13, 128)).astype(np.float32)
    wt = (rng.standard_normal((64, 5, 5, 128)) / np.sqrt(6.25 * 128)).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    beta = (1.0 + rng.uniform(0, .5, 64)).astype(np.float32)
    gamma = (0.1 * np.eye(64) + rng.uniform(0, .05, (64, 64))).astype(np.float32)
    want = oracle.conv2d(x, wt, b, mode=abi.MODE_TCONV, stride=2, gdn=(beta, gamma, True))
    got = ops.conv2d(T(x, cuda), T(wt, cuda), T(b, cuda), mode=abi.MODE_TCONV, stride=2, gdn=(T(beta, cuda), T(gamma, cuda), True))
    assert np.array_equal(got.cpu().numpy(), want)
