"""The bf16x3 precision MODE of the wide convolutions (include/aivc_hip.h: aivc_conv_params.precision; SURVEY.md section 7
"expose precision as a mode and report parity per mode").  Never the default and never the headline: fp32 operands split
exactly into three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulation.  What is checked per mode:

  * every covered layer shape against an fp64 evaluation of the same layer: the mode's error next to the fp32 contract's
    (both are summation-order noise of an fp32 accumulator; the mode must stay within a small factor of the contract);
  * the reference's own outputs at the hot-path widths (tests/golden/wide_*.npz), same bound as the fp32 kernels';
  * the codec in this mode: encoder and decoder agree bit for bit on one GPU (closed loop, clean range-decoder bit count);
  * the reconstruction of the SAME latents by the two modes: 8-bit planes within +-1 LSB;
  * shapes the mode does not cover run the fp32 contract unchanged (bit identical to the default)."""
import numpy as np
import pytest
import torch

from aivc_amd import abi

pytestmark = pytest.mark.gpu

LAYERS = [  # mode, k, stride, pad, c_in, c_out, h, w, fused gdn (0 / 1 / 2), residual
    (abi.MODE_CONV, 5, 2, 2, 64, 128, 68, 120, 1, False),   # the dominant layer: 5x5 s2 64 -> 128 + GDN
    (abi.MODE_CONV, 3, 1, 1, 128, 128, 34, 60, 0, True),    # residual block 3x3
    (abi.MODE_CONV, 5, 2, 2, 128, 64, 34, 60, 0, False),    # last analysis conv, c_out 64 (256x64 tile)
    (abi.MODE_CONV, 3, 1, 1, 64, 128, 17, 30, 0, False),    # 3x3 from 64 channels
    (abi.MODE_TCONV, 5, 2, 0, 128, 128, 17, 30, 2, False),  # transposed 5x5 + inverse GDN
    (abi.MODE_TCONV, 3, 2, 0, 128, 64, 19, 23, 0, False),   # transposed 3x3 to 64 channels, odd sizes
]


def _torch_fp64(mode, x, w, b, stride, pad, k):
    """fp64 evaluation of the layer on the CPU (NHWC in / out): conv with replicate padding, or ConvTranspose2d k, s 2,
    pad (1 + k) / 2 - 1, output_padding 1 (src/layers/misc/custom_conv_layers.py:129-253)"""
    import torch.nn.functional as F
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    wt = torch.from_numpy(w).double()  # OHWI
    if mode == abi.MODE_CONV:
        xp = F.pad(xt, (pad, pad, pad, pad), mode='replicate') if pad else xt
        y = F.conv2d(xp, wt.permute(0, 3, 1, 2), torch.from_numpy(b).double(), stride=stride)
    else:
        y = F.conv_transpose2d(xt, wt.permute(3, 0, 1, 2), torch.from_numpy(b).double(), stride=2,
                               padding=int((1 + k) / 2 - 1), output_padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('case', LAYERS)
def test_layer_error_against_fp64(case, cuda):
    from aivc_amd import ops
    mode, k, s, pad, ci, co, h, w, gdn, use_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((2, h, w, ci), dtype=np.float32)
    wt = (rng.standard_normal((co, k, k, ci), dtype=np.float32) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.standard_normal(co, dtype=np.float32)
    ref = _torch_fp64(mode, x, wt, b, s, pad, k)
    g = None
    if gdn:
        beta = (np.abs(rng.standard_normal(co)) + 0.5).astype(np.float32)
        gamma = (np.abs(rng.standard_normal((co, co))) * 0.02).astype(np.float32)
        nrm = torch.sqrt(torch.from_numpy(beta).double() + (ref * ref) @ torch.from_numpy(gamma).double().t())
        ref = ref * nrm if gdn == 2 else ref / nrm
        g = (torch.from_numpy(beta).to(cuda), torch.from_numpy(gamma).to(cuda), gdn == 2)
    res = None
    if use_res:
        res = rng.standard_normal(tuple(ref.shape), dtype=np.float32)
        ref = ref + torch.from_numpy(res).double()
    ref = ref.numpy()
    scale = np.sqrt(np.mean(ref * ref))
    err = {}
    for name in ('fp32', 'bf16x3'):
        prev = ops.set_precision(name)
        try:
            y = ops.conv2d(torch.from_numpy(x).to(cuda), torch.from_numpy(wt).to(cuda), torch.from_numpy(b).to(cuda), mode=mode,
                           stride=s, pad=pad, gdn=g, res=None if res is None else torch.from_numpy(res).to(cuda))
        finally:
            ops.set_precision(prev)
        d = y.cpu().numpy().astype(np.float64) - ref
        err[name] = (float(np.abs(d).max() / scale), float(np.sqrt(np.mean(d * d)) / scale))
    print('\n%s k%d s%d %d->%d gdn%d: fp32 max %.2e rms %.2e | bf16x3 max %.2e rms %.2e' % (
        'tconv' if mode == abi.MODE_TCONV else 'conv', k, s, ci, co, gdn, *err['fp32'], *err['bf16x3']))
    # fp32 accumulation noise for K up to 3200 terms is ~1e-7 rms of the output scale; the mode may not be worse than a
    # few times the contract's own distance from the fp64 value
    assert err['bf16x3'][1] <= max(4.0 * err['fp32'][1], 2e-7), err
    assert err['bf16x3'][0] <= max(4.0 * err['fp32'][0], 2e-6), err


def test_mode_takes_the_covered_shapes_only(cuda):
    """aivc_conv2d_variant says which kernel a launch takes: 1000 + code for the precision mode; a layer the mode does not
    cover (thin output, image layer, c_out 32, fused 1x1 tail) runs the fp32 contract: bit identical to the default"""
    from aivc_amd import ops
    from aivc_amd._lib import load
    import ctypes as C
    rng = np.random.default_rng(3)

    def variant(mode, k, s, pad, ci, co, h, w, prec):
        ho, wo = abi.conv_out_size(mode, h, w, k, s, pad)
        x = torch.zeros((1, h, w, ci), device=cuda)
        wt = torch.zeros((co, k, k, ci), device=cuda)
        y = torch.zeros((1, ho, wo, co), device=cuda)
        p = abi.ConvParams(mode, k, s, pad, 1, h, w, ci, ho, wo, co, 0, 0, abi.ALGO_AUTO, 0, 0, x.data_ptr(), wt.data_ptr(), None,
                           None, None, y.data_ptr(), None, None)
        p.precision = prec
        return load()['aivc_conv2d_variant'](C.byref(p))
    assert variant(abi.MODE_CONV, 3, 1, 1, 128, 128, 34, 60, abi.PREC_BF16X3) == 1100
    assert variant(abi.MODE_CONV, 5, 2, 2, 128, 64, 34, 60, abi.PREC_BF16X3) == 1102
    assert variant(abi.MODE_TCONV, 5, 2, 0, 128, 128, 17, 30, abi.PREC_BF16X3) == 1110
    assert variant(abi.MODE_CONV, 3, 1, 1, 128, 128, 34, 60, abi.PREC_FP32) < 1000
    for shape in ((abi.MODE_CONV, 3, 1, 1, 128, 32, 20, 20), (abi.MODE_CONV, 5, 2, 2, 8, 64, 40, 40), (abi.MODE_TCONV, 5, 2, 0, 64, 3, 20, 20),
                  (abi.MODE_CONV, 1, 1, 0, 128, 64, 20, 20)):  # (1x1: reduction too short for the mode's tiles)
        assert variant(*shape, abi.PREC_BF16X3) < 1000
        mode, k, s, pad, ci, co, h, w = shape
        x = torch.from_numpy(rng.standard_normal((1, h, w, ci), dtype=np.float32)).to(cuda)
        wt = torch.from_numpy(rng.standard_normal((co, k, k, ci), dtype=np.float32) * 0.05).to(cuda)
        a = ops.conv2d(x, wt, None, mode=mode, stride=s, pad=pad)
        prev = ops.set_precision('bf16x3')
        try:
            b = ops.conv2d(x, wt, None, mode=mode, stride=s, pad=pad)
        finally:
            ops.set_precision(prev)
        assert torch.equal(a, b)


def test_wide_reference_fixtures_in_bf16x3(cuda, golden):
    """the reference's own outputs at the hot-path widths (tests/golden/wide_*.npz): the mode meets the bound the fp32
    kernels are held to (2e-5 relative to max(1, |y|)); the error of each case is printed next to the fp32 kernels'"""
    from test_wide_golden import NAMES, _build, _run_gpu
    from aivc_amd import ops
    worst = {}
    took = set()
    for name in NAMES:
        g = golden('wide_' + name)
        m, x, _ = _build(name)
        errs = []
        for mode in ('fp32', 'bf16x3'):
            prev = ops.set_precision(mode)
            try:
                y, variants = _run_gpu(m, x, cuda)
            finally:
                ops.set_precision(prev)
            errs.append(float((np.abs(y - g['y']) / np.maximum(1.0, np.abs(g['y']))).max()))
            if mode == 'bf16x3':
                took |= {v for v in variants if v >= 1000}
        worst[name] = errs
    print('\nmax relative error vs the reference outputs, fp32 | bf16x3:')
    for k, (e0, e1) in worst.items():
        print('  %-22s %.2e | %.2e' % (k, e0, e1))
    assert took, 'no launch of these cases took the precision mode'
    assert max(e[1] for e in worst.values()) <= 2e-5, worst


def test_codec_closed_loop_and_cross_mode_reconstruction(cuda):
    """default-width model, 416x240, 9 frames RA: in bf16x3 mode the decoder reproduces the encoder's reconstruction bit
    for bit and every range-decoder stream ends where its payload ends; the SAME latents reconstructed by the two modes
    give 8-bit planes within +-1 LSB"""
    from aivc_amd import ops, synth
    from aivc_amd.codec import FrameCodec
    from aivc_amd.models import arch
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=cuda)
    synth.calibrate_operating_point(model, cuda)
    frames = synth.to_device_frames(synth.synthetic_video(416, 240, 9, seed=9), cuda)
    fc = FrameCodec(model)
    prev = ops.set_precision('bf16x3')
    try:
        with torch.no_grad():
            enc = fc.encode_video(frames, '1_GOP_8')
            blob = fc.assemble_video(enc)
            dec, _, _, _ = fc.decode_video(blob, cuda)
        assert fc.stream_errors() == []
        rec = [r for g in enc['recs'] for r in g][:len(dec)]
        for i, (d, e) in enumerate(zip(dec, rec)):
            for k in 'yuv':
                assert torch.equal(d[k], e[k]), (i, k)
    finally:
        ops.set_precision(prev)
    # the same I-frame latents through the synthesis of both modes
    with torch.no_grad():
        out = fc.encode_batch([frames[0]], [None], [None], 0)
        from aivc_amd.real_life.bitstream import finalize_frames
        fb = finalize_frames(out['sections'])
        yh = fc.entropy_decode(fb, 0, out['data_dim'], device=cuda)
        a = fc.synthesise_batch(yh, [None], [None], 0, out['data_dim'])[0]
        prev = ops.set_precision('bf16x3')
        try:
            b = fc.synthesise_batch(yh, [None], [None], 0, out['data_dim'])[0]
        finally:
            ops.set_precision(prev)
    diff = {k: (a[k].int() - b[k].int()).abs() for k in 'yuv'}
    frac = sum(int((v > 0).sum()) for v in diff.values()) / sum(v.numel() for v in diff.values())
    print('\nsame latents, fp32 vs bf16x3 synthesis: %.4f %% of the 8-bit samples differ (by 1 LSB)' % (100 * frac))
    assert max(int(v.max()) for v in diff.values()) <= 1


def test_weight_split_equals_the_oracle(cuda, oracle):
    """aivc_split_weights_bf16x3: three bf16 terms per weight, x == h + m + l exactly, laid out [row][tile of 32][term][32]
    -- HIP == the CPU restatement bit for bit, and the terms really sum to the weight"""
    from aivc_amd import ops
    rng = np.random.default_rng(7)
    w = (rng.standard_normal((24, 3, 3, 32), dtype=np.float32) * np.exp(rng.uniform(-20, 20, (24, 1, 1, 1)))).astype(np.float32)
    got = ops.split_weights_bf16x3(torch.from_numpy(w).to(cuda)).cpu().numpy().view(np.uint16)
    want = np.empty(w.size * 3, np.uint16)
    lib = oracle.lib()
    assert lib['aivc_split_weights_bf16x3'](w.ctypes.data, 24, 288, want.ctypes.data, None) == 0
    np.testing.assert_array_equal(got, want)
    t = (got.reshape(24, 9, 3, 32).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    np.testing.assert_array_equal(t.sum(axis=2).reshape(24, 288), w.reshape(24, 288).astype(np.float64))


@pytest.mark.parametrize('case', LAYERS + [(abi.MODE_CONV, 5, 2, 2, 64, 64, 40, 52, 1, False),
                                          (abi.MODE_CONV, 3, 1, 1, 128, 256, 9, 70, 0, False),
                                          (abi.MODE_TCONV, 5, 2, 0, 128, 64, 21, 37, 1, False)])
def test_split_weights_ahead_of_the_launch_change_no_bit(case, cuda):
    """aivc_conv_params.w_bf16x3: the kernels that read the weights' three terms from memory return exactly what the
    kernels that split the fragments in their K loop return (partial tiles, both tile shapes, fused (I)GDN, residual)"""
    from aivc_amd import ops
    mode, k, s, pad, ci, co, h, w, gdn, use_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 1)
    x = torch.from_numpy(rng.standard_normal((3, h, w, ci), dtype=np.float32)).to(cuda)
    wt = torch.from_numpy((rng.standard_normal((co, k, k, ci), dtype=np.float32) / np.sqrt(k * k * ci)).astype(np.float32)).to(cuda)
    b = torch.from_numpy(rng.standard_normal(co, dtype=np.float32)).to(cuda)
    g = None
    if gdn:
        g = (torch.from_numpy((np.abs(rng.standard_normal(co)) + 0.5).astype(np.float32)).to(cuda),
             torch.from_numpy((np.abs(rng.standard_normal((co, co))) * 0.02).astype(np.float32)).to(cuda), gdn == 2)
    ho, wo = abi.conv_out_size(mode, h, w, k, s, pad)
    res = torch.from_numpy(rng.standard_normal((3, ho, wo, co), dtype=np.float32)).to(cuda) if use_res else None
    prev = ops.set_precision('bf16x3')
    try:
        out = {}
        for ahead in (False, True):
            ops.PRESPLIT_WEIGHTS = ahead
            out[ahead] = ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, gdn=g, res=res)
    finally:
        ops.PRESPLIT_WEIGHTS = True
        ops.set_precision(prev)
    assert torch.equal(out[False], out[True])
    prev = ops.set_precision('fp32')
    try:
        taps = (k * k + 3) // 4 if mode == abi.MODE_TCONV else k * k
        if taps * ci >= 512:  # (shorter reductions stay on the fp32 kernels)
            assert not torch.equal(ops.conv2d(x, wt, b, mode=mode, stride=s, pad=pad, gdn=g, res=res), out[True])  # (the mode did run)
    finally:
        ops.set_precision(prev)


def test_fused_tail_layer_in_the_mode(cuda):
    """The bottleneck blocks' 3x3 64 -> 64 + fused 1x1 64 -> 128 (+ residual): the 3x3 runs in the mode, the 1x1 GEMM stays
    fp32 -- within summation-order noise of the contract's result, weights split ahead or in the loop: the same bits"""
    from aivc_amd import ops
    rng = np.random.default_rng(11)
    n, h, w = 3, 37, 53
    x = torch.from_numpy(rng.standard_normal((n, h, w, 64), dtype=np.float32)).to(cuda)
    w1 = torch.from_numpy((rng.standard_normal((64, 3, 3, 64), dtype=np.float32) / 24.0).astype(np.float32)).to(cuda)
    b1 = torch.from_numpy(rng.standard_normal(64, dtype=np.float32)).to(cuda)
    w3 = torch.from_numpy((rng.standard_normal((128, 1, 1, 64), dtype=np.float32) / 8.0).astype(np.float32)).to(cuda)
    b3 = torch.from_numpy(rng.standard_normal(128, dtype=np.float32)).to(cuda)
    res = torch.from_numpy(rng.standard_normal((n, h, w, 128), dtype=np.float32)).to(cuda)

    def run():
        return ops.conv2d(x, w1, b1, stride=1, pad=1, act1=abi.ACT_RELU, res=res, tail=(w3, b3))
    want = run()
    prev = ops.set_precision('bf16x3')
    try:
        ops.PRESPLIT_WEIGHTS = False
        in_loop = run()
        ops.PRESPLIT_WEIGHTS = True
        ahead = run()
    finally:
        ops.PRESPLIT_WEIGHTS = True
        ops.set_precision(prev)
    assert torch.equal(in_loop, ahead)
    assert not torch.equal(ahead, want)  # (the mode did run)
    d = (ahead.double() - want.double()).abs().max().item() / want.double().pow(2).mean().sqrt().item()
    print('\nfused tail: bf16x3 vs fp32 contract, max |d| / rms = %.2e' % d)
    assert d < 2e-5


@pytest.mark.parametrize('case', ['decoder_mid_gop8', 'decoder_big_gop8', 'decoder_ra', 'decoder_b_mid_gop8'])
def test_reference_run_decoder_fixtures_in_bf16x3(case, cuda, golden, monkeypatch):
    """The reference-run decoder fixtures (tests/test_decoder_golden.py: `.bin` files written by the reference's own
    ArithmeticCoder.encode / container writers, decoded by its decode_one_video to PNG planes) decoded by the HIP path
    IN THE MODE: planes within +-1 LSB of the reference's, inside the pixel budget of the fp32 contract's test, every
    section's bit count clean.  decoder_mid_gop8 is the mid-width model (c_in % 32 == 0 on every layer behind the image
    layers): its wide conv / transposed conv layers must actually TAKE the mode (variant >= 1000), otherwise the test
    proves nothing about it; the small-width cases run the fp32 contract unchanged (the mode covers no shape of theirs)
    and must still pass with the switch on."""
    import test_decoder_golden as tdg
    from aivc_amd import ops
    g = golden(case)
    m = tdg._meta(g)
    model = tdg._model(golden, case, cuda)
    tdg._teach_sigma(model, g, m, monkeypatch)
    fc = model.frame_codec()
    prev = ops.set_precision('bf16x3')
    ops.PROFILE = []
    try:
        with torch.no_grad():
            dec, data_dim, first, last = fc.decode_video(np.asarray(g['video_file']).tobytes(), cuda)
        torch.cuda.synchronize()
        taken = sorted({p[0] for p in ops.PROFILE if p[0] >= 1000})
        n_mode = sum(1 for p in ops.PROFILE if p[0] >= 1000)
        n_all = len(ops.PROFILE)
    finally:
        ops.PROFILE = None
        ops.set_precision(prev)
    want = tdg._frames(g, m, 'dec')
    n_off = 0
    for d, w in zip(dec, want):
        for k in 'yuv':
            diff = np.abs(d[k][0].cpu().numpy().astype(np.int32) - w[k].astype(np.int32))
            assert diff.max() <= 1, (case, k)
            n_off += int((diff != 0).sum())
    print('\n%s in bf16x3: %d of %d conv launches in the mode (variants %s), %d samples off by 1 LSB (budget %d)'
          % (case, n_mode, n_all, taken, n_off, tdg._pixel_budget(m, want)))
    assert n_off <= tdg._pixel_budget(m, want)
    assert fc.stream_errors() == []
    if case == 'decoder_mid_gop8':
        assert n_mode > 0, 'the mid-width decoder ran no launch in the precision mode'
