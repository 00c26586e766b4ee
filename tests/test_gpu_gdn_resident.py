"""csrc/gdn.hip -- the stand-alone (I)GDN kernel with gamma resident in registers: the bits of the CPU oracle and of the
generic kernel's GDN-mode launch (aivc_conv2d with algo = AIVC_ALGO_MFMA) on 64 / 128 channels, partial tiles,
more tiles than workgroups, residual, inverse, operands outside the lean sqrt / division range."""
import ctypes as C

import numpy as np
import pytest
import torch

from aivc_amd import abi

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _variant(c, n, h, w, inverse, dev):
    from aivc_amd import _lib
    x = torch.zeros((n, h, w, c), device=dev)
    g = torch.zeros((c, c), device=dev)
    b = torch.ones(c, device=dev)
    p = abi.ConvParams(abi.MODE_IGDN if inverse else abi.MODE_GDN, 1, 1, 0, n, h, w, c, h, w, c, 0, 0, abi.ALGO_AUTO, 0, 0,
                       x.data_ptr(), g.data_ptr(), b.data_ptr(), None, None, x.data_ptr(), None, None)
    return _lib.load()['aivc_conv2d_variant'](C.byref(p))


CASES = [
    # c, n, h, w, inverse, res
    (128, 1, 8, 8, False, False),      # one whole tile
    (128, 1, 5, 7, False, False),      # one partial tile (35 of 64 rows)
    (128, 2, 33, 31, False, True),     # 2046 pixels: 31 whole tiles + 62 rows
    (128, 2, 33, 31, True, False),
    (128, 1, 9, 13, True, True),
    (64, 1, 33, 31, False, False),
    (64, 2, 17, 19, True, True),
    (64, 1, 3, 5, False, True),
    (128, 3, 136, 120, False, False),  # 765 tiles: more than the 512 persistent workgroups of a 256-CU part
    (128, 3, 136, 120, True, True),
]


@pytest.mark.parametrize('case', CASES)
def test_resident_gdn_bit_exact(case, oracle, cuda):
    from aivc_amd import ops
    c, n, h, w, inv, use_res = case
    assert _variant(c, n, h, w, inv, cuda) == 400  # the kernel under test takes the launch
    rng = np.random.default_rng(c * 1000 + n * 100 + h + w + (5 if inv else 0))
    x = rng.standard_normal((n, h, w, c), dtype=np.float32)
    beta = (np.abs(rng.standard_normal(c)) + 0.2).astype(np.float32)
    gamma = (np.abs(rng.standard_normal((c, c))) * 0.05).astype(np.float32)
    res = rng.standard_normal((n, h, w, c), dtype=np.float32) if use_res else None
    rt = None if res is None else T(res, cuda)
    got = ops.gdn(T(x, cuda), T(beta, cuda), T(gamma, cuda), inverse=inv, res=rt)
    generic = ops.gdn(T(x, cuda), T(beta, cuda), T(gamma, cuda), inverse=inv, res=rt, algo=abi.ALGO_MFMA)
    assert torch.equal(got, generic)
    if n * h * w <= 4096:  # (the CPU oracle at the large sizes: the generic kernel stands in, itself pinned above and in test_gpu_ops)
        want = oracle.gdn(x, beta, gamma, inverse=inv, res=res)
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('inv', [False, True])
@pytest.mark.parametrize('kind', ['zeros', 'tiny', 'huge', 'mixed'])
def test_resident_gdn_outside_the_lean_range(kind, inv, oracle, cuda):
    """operands outside [2^-60, 2^60] (and exact zeros): the wavefront takes the compiler's full IEEE sequences"""
    from aivc_amd import ops
    c, n, h, w = 128, 1, 9, 15
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n, h, w, c), dtype=np.float32)
    beta = (np.abs(rng.standard_normal(c)) + 0.2).astype(np.float32)
    gamma = (np.abs(rng.standard_normal((c, c))) * 0.05).astype(np.float32)
    if kind == 'zeros':
        x[:, ::2] = 0.0
    elif kind == 'tiny':
        x *= np.float32(2.0 ** -70)
        beta *= np.float32(2.0 ** -100)
    elif kind == 'huge':
        x *= np.float32(2.0 ** 40)
    else:
        x[0, 3, 4, 5] = 0.0
        x[0, 7, 1, 99] = np.float32(2.0 ** 50)
    want = oracle.gdn(x, beta, gamma, inverse=inv)
    got = ops.gdn(T(x, cuda), T(beta, cuda), T(gamma, cuda), inverse=inv)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_unsupported_shapes_fall_back(cuda):
    """96 / 192 channels: the generic kernel keeps the launch"""
    assert _variant(96, 1, 9, 9, False, cuda) != 400
    assert _variant(192, 1, 9, 9, False, cuda) != 400
    assert _variant(8, 1, 9, 9, False, cuda) != 400
