"""aivc_amd -- MI355X-native hot path of the AIVC learned video codec.

The sub-packages mirror the reference's module tree (``layers``, ``func_util``, ``models``,
``real_life``, ``model_mngt``) so that its pickled ``.pt`` models and its CLI keep working; the
arithmetic underneath runs in hand-written HIP kernels (aivc_amd/csrc, C ABI in include/aivc_hip.h).
"""
import importlib
import os
import sys

# The codec runs its entropy stages on up to 8 side streams next to the transforms' stream (codec.py).  The HIP
# runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a
# queue serialise: with 4, the entropy stage of a dependency level queues behind an earlier level's and the main
# stream waits for it (measured: decode 190 -> 195 fps with 8).  Read by the runtime when it initialises, i.e. at the
# first device call -- importing this package before touching the GPU is enough; an explicit setting wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'

_ALIASED = ('layers', 'func_util', 'models', 'real_life', 'model_mngt')


def install_aliases():
    """Register the reference's top-level module names (``layers.misc.custom_conv_layers`` ...) as
    aliases of this package's modules, so ``torch.load`` of a reference full-module pickle resolves
    its classes here.  Idempotent."""
    import pkgutil
    for top in _ALIASED:
        pkg = importlib.import_module('aivc_amd.' + top)
        sys.modules.setdefault(top, pkg)
        for m in pkgutil.walk_packages(pkg.__path__, 'aivc_amd.' + top + '.'):
            mod = importlib.import_module(m.name)
            sys.modules.setdefault(m.name[len('aivc_amd.'):], mod)
