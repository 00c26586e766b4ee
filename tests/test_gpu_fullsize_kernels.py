"""Full-size (BASELINE configs[3] / [4] shapes) checks of the fused kernels through size-independent properties: a
fused launch equals the launches it replaces, bit for bit, where the oracle would take minutes; the range coder
round-trips a 4K-sized latent with out-of-window symbols."""
import numpy as np
import pytest
import torch

from aivc_amd import abi

pytestmark = pytest.mark.gpu


def test_fused_tail_full_size_equals_two_launches(cuda):
    """the attention bottleneck at 1080p/8 (135 x 240), 4 frames: one launch == 3x3 conv then 1x1 conv (+res, leaky)"""
    from aivc_amd import ops
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn((4, 135, 240, 64), generator=g).to(cuda)
    res = torch.randn((4, 135, 240, 128), generator=g).to(cuda)
    w = (torch.randn((64, 3, 3, 64), generator=g) / 24.0).to(cuda)
    b = torch.randn(64, generator=g).to(cuda)
    w3 = (torch.randn((128, 1, 1, 64), generator=g) / 8.0).to(cuda)
    b3 = torch.randn(128, generator=g).to(cuda)
    fused = ops.conv2d(x, w, b, stride=1, pad=1, act1=abi.ACT_LEAKY, act2=abi.ACT_LEAKY, res=res, tail=(w3, b3))
    t = ops.conv2d(x, w, b, stride=1, pad=1, act1=abi.ACT_LEAKY)
    two = ops.conv2d(t, w3, b3, res=res, act2=abi.ACT_LEAKY)
    assert torch.equal(fused, two)


def test_fused_gdn_residual_full_size_equals_two_launches(cuda):
    """closing conv of a residual block at 1080p/4: conv + fused GDN + residual == conv, then GDN launch with residual"""
    from aivc_amd import ops
    g = torch.Generator(device='cpu').manual_seed(4)
    x = torch.randn((2, 270, 480, 128), generator=g).to(cuda)
    res = torch.randn((2, 270, 480, 128), generator=g).to(cuda)
    w = (torch.randn((128, 3, 3, 128), generator=g) / 34.0).to(cuda)
    b = torch.randn(128, generator=g).to(cuda)
    beta = (torch.rand(128, generator=g) + 0.5).to(cuda)
    gamma = (torch.rand((128, 128), generator=g) * 0.01).to(cuda)
    for inv in (False, True):
        fused = ops.conv2d(x, w, b, stride=1, pad=1, res=res, gdn=(beta, gamma, inv))
        two = ops.gdn(ops.conv2d(x, w, b, stride=1, pad=1), beta, gamma, inverse=inv, res=res)
        assert torch.equal(fused, two)


@pytest.mark.parametrize('n_img', [1, 2, 3])
def test_conv_images_full_size_equals_pack_then_conv(n_img, cuda, monkeypatch):
    """first analysis layer at 1080p from 8-bit planes: aivc_conv_images == aivc_pack_images + aivc_conv2d"""
    from aivc_amd import ops
    monkeypatch.setattr(ops, '_CONV_IMAGES_MAX', 3)
    g = torch.Generator(device='cpu').manual_seed(10 + n_img)
    n, h, w = 2, 1080, 1920

    def planes():
        return {'y': torch.randint(0, 256, (n, h, w), generator=g, dtype=torch.uint8).to(cuda),
                'u': torch.randint(0, 256, (n, h // 2, w // 2), generator=g, dtype=torch.uint8).to(cuda),
                'v': torch.randint(0, 256, (n, h // 2, w // 2), generator=g, dtype=torch.uint8).to(cuda)}
    parts = [planes() for _ in range(n_img)]
    wt = torch.zeros((64, 5, 5, 4 * n_img))
    for i in range(n_img):
        wt[..., 4 * i:4 * i + 3] = torch.randn((64, 5, 5, 3), generator=g) / (75 * n_img) ** 0.5
    wt = wt.to(cuda)
    b = torch.randn(64, generator=g).to(cuda)
    gd = ((torch.rand(64, generator=g) + 0.5).to(cuda), (torch.rand((64, 64), generator=g) * 0.01).to(cuda), False)
    stack = ops.ImageStack(parts, h, w, cuda)
    direct = ops.conv2d(stack, wt, b, stride=2, pad=2, gdn=gd)
    assert stack._packed is None
    two = ops.conv2d(ops.pack_images(parts, h, w, cuda), wt, b, stride=2, pad=2, gdn=gd)
    assert torch.equal(direct, two)


def test_range_coder_round_trip_4k_latent(cuda):
    """one stream of a 2160p latent (136 x 240 x 64 = 2.09 M symbols), sigma sweep incl. symbols outside the decoder's
    fast window: encode -> decode returns the symbols"""
    from aivc_amd import ops
    g = torch.Generator(device='cpu').manual_seed(21)
    h, w, c = 136, 240, 64
    sigma = torch.exp(torch.rand((1, h, w, c), generator=g) * 6.0 - 2.0).to(cuda)      # 0.13 .. 55
    q = torch.round(torch.randn((1, h, w, c), generator=g).to(cuda) * sigma).clamp_(-256, 255).to(torch.int16)
    maps = list(range(c))
    bounds = ops.laplace_bounds(sigma, q, maps)
    out, lens, offs = ops.range_encode([bounds])
    nbytes = int(lens.cpu()[0])
    payload = out[offs[0][0]:offs[0][0] + nbytes].cpu().numpy().tobytes()
    assert 0 < nbytes < 2 * h * w * c
    win, sp = ops.laplace_cdf_windows(sigma, maps)
    dec = ops.range_decode([payload], win, [0], [h * w * c], [0], sigma_pos=sp)[0]
    want = (q.permute(3, 0, 1, 2).reshape(-1).to(torch.int32) + 256).to(torch.int16)
    assert (q.abs() > 40).any(), 'the sweep must leave the 64-entry window'
    assert torch.equal(dec.view(torch.int16).reshape(-1), want)


def test_batch_beyond_4gb_goes_out_as_sub_batches(cuda):
    """the LDS-DMA loader addresses its input with 32-bit BYTE offsets: a batch of more than 4 GB (34 half-resolution
    64-channel maps of a 1080p frame = 4.5 GB) is split into sub-batch launches inside aivc_conv2d -- same bits as the
    caller splitting it, fused GDN and the transposed conv included"""
    from aivc_amd import ops
    g = torch.Generator(device='cpu').manual_seed(21)
    x = torch.randn((34, 540, 960, 64), generator=g).to(cuda)
    w = (torch.randn((128, 5, 5, 64), generator=g) / 40.0).to(cuda)
    b = torch.randn(128, generator=g).to(cuda)
    beta = (torch.rand(128, generator=g) + 0.5).to(cuda)
    gamma = (torch.rand((128, 128), generator=g) * 0.01).to(cuda)
    assert x.numel() * 4 > 2 ** 32
    full = ops.conv2d(x, w, b, stride=2, pad=2, gdn=(beta, gamma, False))
    for lo, hi in ((0, 17), (17, 34)):
        assert torch.equal(full[lo:hi], ops.conv2d(x[lo:hi].contiguous(), w, b, stride=2, pad=2, gdn=(beta, gamma, False)))
    del full
    prev = ops.set_precision('bf16x3')  # the precision mode's launches split the same way (its own kernels)
    try:
        full = ops.conv2d(x, w, b, stride=2, pad=2, gdn=(beta, gamma, False))
        assert torch.equal(full[30:34], ops.conv2d(x[30:34].contiguous(), w, b, stride=2, pad=2, gdn=(beta, gamma, False)))
        del full
    finally:
        ops.set_precision(prev)
    wt = (torch.randn((32, 3, 3, 64), generator=g) / 24.0).to(cuda)
    bt = torch.randn(32, generator=g).to(cuda)
    full = ops.conv2d(x, wt, bt, mode=abi.MODE_TCONV, stride=2, act1=abi.ACT_LEAKY)
    assert torch.equal(full[30:34], ops.conv2d(x[30:34].contiguous(), wt, bt, mode=abi.MODE_TCONV, stride=2, act1=abi.ACT_LEAKY))
