"""CPU oracle: the fused 1x1 tail of aivc_conv2d (include/aivc_hip.h) is bit-identical to the two convolutions it
replaces (the bottleneck blocks of the attention module, src/layers/misc/attention.py:22-42)."""
import numpy as np
import pytest

from aivc_amd import abi


@pytest.mark.parametrize('case', [
    # k, stride, cin, c_mid, c_tail, n, h, w, act1, act2, res
    (3, 1, 8, 8, 16, 2, 9, 11, abi.ACT_LEAKY, abi.ACT_LEAKY, True),
    (3, 1, 64, 64, 128, 1, 6, 7, abi.ACT_LEAKY, abi.ACT_LEAKY, True),
    (5, 2, 4, 12, 8, 1, 13, 9, abi.ACT_RELU, abi.ACT_NONE, False),
    (1, 1, 16, 20, 4, 2, 5, 5, abi.ACT_NONE, abi.ACT_RELU, True),
])
def test_fused_tail_equals_two_convs(case, oracle):
    k, s, ci, cm, ct, n, h, w, a1, a2, use_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((n, h, w, ci), dtype=np.float32)
    wt = (rng.standard_normal((cm, k, k, ci), dtype=np.float32) / np.sqrt(k * k * ci)).astype(np.float32)
    b1 = rng.standard_normal(cm, dtype=np.float32)
    w3 = (rng.standard_normal((ct, 1, 1, cm), dtype=np.float32) / np.sqrt(cm)).astype(np.float32)
    b3 = rng.standard_normal(ct, dtype=np.float32)
    ho, wo = abi.conv_out_size(abi.MODE_CONV, h, w, k, s, k // 2)
    res = rng.standard_normal((n, ho, wo, ct), dtype=np.float32) if use_res else None
    t = oracle.conv2d(x, wt, b1, stride=s, pad=k // 2, act1=a1)
    two = oracle.conv2d(t, w3, b3, res=res, act2=a2)
    fused = oracle.conv2d(x, wt, b1, stride=s, pad=k // 2, act1=a1, act2=a2, res=res, tail=(w3, b3))
    assert fused.shape == (n, ho, wo, ct)
    np.testing.assert_array_equal(fused, two)


def test_tail_argument_checks(oracle):
    x = np.zeros((1, 4, 4, 4), np.float32)
    w = np.zeros((4, 1, 1, 4), np.float32)
    w3 = np.zeros((8, 1, 1, 4), np.float32)
    with pytest.raises(Exception):  # a gate operand cannot be combined with a tail
        oracle.conv2d(x, w, None, mul=np.zeros((1, 4, 4, 8), np.float32), tail=(w3, None))
    with pytest.raises(Exception):  # nor a fused gdn
        oracle.conv2d(x, w, None, gdn=(np.ones(4, np.float32), np.zeros((4, 4), np.float32), False), tail=(w3, None))


def test_conv_images_twin_equals_pack_then_conv(oracle):
    """aivc_conv_images_ref is by definition aivc_conv2d_ref over aivc_pack_images_ref"""
    rng = np.random.default_rng(11)
    n, h, w = 1, 11, 14
    hc, wc = (h + 1) // 2, (w + 1) // 2
    a = {'y': rng.integers(0, 256, (n, h, w), dtype=np.uint8), 'u': rng.integers(0, 256, (n, hc, wc), dtype=np.uint8),
         'v': rng.integers(0, 256, (n, hc, wc), dtype=np.uint8)}
    f = rng.standard_normal((n, h, w, 3)).astype(np.float32)
    wt = np.zeros((8, 5, 5, 8), np.float32)
    wt[..., :3] = rng.standard_normal((8, 5, 5, 3)).astype(np.float32)
    wt[..., 4:7] = rng.standard_normal((8, 5, 5, 3)).astype(np.float32)
    bias = rng.standard_normal(8, dtype=np.float32)
    want = oracle.conv2d(oracle.pack_images([a, f], h, w), wt, bias, stride=2, pad=2, act1=abi.ACT_RELU)
    np.testing.assert_array_equal(oracle.conv_images([a, f], h, w, wt, bias, act1=abi.ACT_RELU), want)
