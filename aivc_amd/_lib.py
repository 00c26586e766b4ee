"""Loader of the native HIP library (aivc_amd/lib/libaivc_hip.so, built in-tree by
__graft_entry__.build()).  There is NO fallback: if the library is missing or does not export the
whole ABI of include/aivc_hip.h the import of any compute path fails loudly."""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# AIVC_HIP_LIB: kernel-tuning aid (tools/), points at an alternative build of the same HIP library
LIB_PATH = os.environ.get('AIVC_HIP_LIB') or os.path.join(_HERE, 'lib', 'libaivc_hip.so')

_lib = None
_fns = None


class AivcNativeError(RuntimeError):
    pass


def load():
    """Return {name: ctypes function}.  Raises AivcNativeError when the HIP library is absent."""
    global _lib, _fns
    if _fns is not None:
        return _fns
    if not os.path.exists(LIB_PATH):
        raise AivcNativeError(
            'aivc_amd: native library %s not found. Build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950). '
            'There is no CPU fallback in the product path.' % LIB_PATH)
    try:
        _lib = C.CDLL(LIB_PATH)
        fns = abi.declare(_lib)
    except (OSError, AttributeError) as e:
        raise AivcNativeError('aivc_amd: cannot load %s: %s' % (LIB_PATH, e))
    err = _lib.aivc_last_error
    err.argtypes = []
    err.restype = C.c_char_p
    fns['aivc_last_error'] = err
    if fns['aivc_abi_version']() != abi.ABI_VERSION:
        raise AivcNativeError('aivc_amd: ABI mismatch between abi.py and %s' % LIB_PATH)
    _fns = fns
    return _fns


def call(name, *args):
    fns = load()
    rc = fns[name](*args)
    if rc != 0:
        detail = fns['aivc_last_error']() or b''
        raise AivcNativeError('%s failed: %s %s' % (name, abi.ERRORS.get(rc, rc), detail.decode()))
