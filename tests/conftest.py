import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from aivc_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _contract_does_not_leak():
    """a test that switches the arithmetic contract / precision mode hands the process back in the default one (a leaked
    'fp32' once made every later test of a session run version 1)"""
    yield
    import sys
    ops = sys.modules.get('aivc_amd.ops')
    if ops is not None:
        default = ops._CONTRACT_NAMES[ops.DEFAULT_CONTRACT]
        leaked = ops.PRECISION != default or ops.WINO_ANY_SIZE
        ops.PRECISION = default
        ops.WINO_ANY_SIZE = False
        assert not leaked, 'the test left ops.PRECISION / ops.WINO_ANY_SIZE changed'
    orc = sys.modules.get('oracle.oracle')
    if orc is not None:
        import os
        from aivc_amd import abi
        d = {'fp32': abi.PREC_FP32, 'fp32w': abi.PREC_FP32_WINO}[os.environ.get('AIVC_CONTRACT', 'fp32w')]
        leaked = orc.PRECISION != d or orc.WINO_ANY_SIZE
        orc.PRECISION = d
        orc.WINO_ANY_SIZE = False
        assert not leaked, 'the test left oracle.PRECISION / oracle.WINO_ANY_SIZE changed'
