"""Class resolution for the reference's full-module pickles (src/model_mngt/model_management.py:346-347 does a plain
torch.load of an nn.Module; upstream saved it with torch.save(model)).

A `.pt` names every class by dotted path.  The layer classes live at paths this package mirrors (`layers.misc.*`,
`layers.ae.*`, ...; aivc_amd.install_aliases()).  The classes of the `models` package are MISSING from the reference
snapshot, so their real dotted paths are unknown until a real pickle is at hand: `resolver()` therefore
first tries the dotted path as written and then falls back to a lookup by CLASS NAME over every class this package
defines, and `pickle_globals()` lists what a file asks for without unpickling anything (tools/inspect_pickle.py).
Names that resolve neither way are reported all at once, before the load starts."""
import functools
import importlib
import io
import pickle
import pickletools
import pkgutil
import sys
import zipfile


def pickle_globals(path_or_bytes):
    """Every (module, name) a pickle would import, in order of first appearance -- read from the GLOBAL /
    STACK_GLOBAL opcodes of `data.pkl` inside a torch zip archive (or of a bare / legacy pickle stream), WITHOUT
    unpickling.  String operands of STACK_GLOBAL are tracked through the memo (BINPUT / MEMOIZE / BINGET)."""
    raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, 'rb').read()
    streams = []
    if zipfile.is_zipfile(io.BytesIO(raw)):
        with zipfile.ZipFile(io.BytesIO(raw)) as zf:
            streams = [zf.read(n) for n in zf.namelist() if n.endswith('data.pkl')]
    else:
        streams = [raw]
    found = []
    for data in streams:
        pos = 0
        while pos < len(data):  # a legacy torch file is several pickles back to back
            memo, stack = {}, []
            try:
                for op, arg, p in pickletools.genops(io.BytesIO(data[pos:])):
                    end = pos + p + 1
                    if op.name == 'GLOBAL':
                        mod, name = arg.split(' ', 1)
                        found.append((mod, name))
                        stack.append(None)
                    elif op.name == 'STACK_GLOBAL':
                        if len(stack) >= 2 and isinstance(stack[-1], str) and isinstance(stack[-2], str):
                            found.append((stack[-2], stack[-1]))
                        del stack[-2:]
                        stack.append(None)
                    elif op.name in ('SHORT_BINUNICODE', 'BINUNICODE', 'BINUNICODE8', 'UNICODE'):
                        stack.append(arg)
                    elif op.name in ('BINPUT', 'LONG_BINPUT', 'PUT'):
                        memo[arg] = stack[-1] if stack else None
                    elif op.name == 'MEMOIZE':
                        memo[len(memo)] = stack[-1] if stack else None
                    elif op.name in ('BINGET', 'LONG_BINGET', 'GET'):
                        stack.append(memo.get(arg))
                    elif op.name == 'STOP':
                        break
                    else:
                        stack.append(None)  # operand tracking only needs the two strings right below STACK_GLOBAL
                        if len(stack) > 64:
                            del stack[:-8]
                pos = end
            except Exception:
                break
    out, seen = [], set()
    for item in found:
        if item not in seen:
            seen.add(item)
            out.append(item)
    return out


_BY_NAME = None
# top-level packages of the reference tree (src/): the only module paths whose classes may be bound by name
_REFERENCE_TOPS = ('models', 'layers', 'func_util', 'real_life', 'model_mngt', 'clic21', 'format_conversion', '__main__')


def classes_by_name():
    """{class name: class} over the mirrored packages of aivc_amd; a name defined twice maps to None (ambiguous)"""
    global _BY_NAME
    if _BY_NAME is None:
        import aivc_amd
        table = {}
        for top in aivc_amd._ALIASED:
            pkg = importlib.import_module('aivc_amd.' + top)
            mods = [pkg] + [importlib.import_module(m.name) for m in pkgutil.walk_packages(pkg.__path__, pkg.__name__ + '.')]
            for mod in mods:
                for k, v in vars(mod).items():
                    if isinstance(v, type) and v.__module__ == mod.__name__:
                        table[k] = None if (k in table and table[k] is not v) else v
        _BY_NAME = table
    return _BY_NAME


def resolve(module, name, log=None):
    """the object a pickle means by (module, name): the dotted path as written (reference module names are aliases of
    this package's), else -- for classes outside torch / numpy / the standard library -- the aivc_amd class of the same
    NAME.  Raises AttributeError when neither exists."""
    import aivc_amd
    aivc_amd.install_aliases()
    try:  # the standard lookup, with pickle's Python-2 name mapping (protocol 2 files say __builtin__, copy_reg ...)
        return pickle.Unpickler(io.BytesIO(b'')).find_class(module, name)
    except (ImportError, AttributeError):
        pass
    if '.' in name:  # a protocol-4 qualified name ('Outer.Inner'): a bare Unpickler is protocol 0 and does not split it
        try:
            return functools.reduce(getattr, name.split('.'), importlib.import_module(module))
        except (ImportError, AttributeError):
            pass
    # by class NAME: only for paths under the reference's own top-level packages (the missing `models` package and
    # whatever upstream called its siblings / a 'src.' prefix) -- an unknown third-party module never binds to a local class
    parts = module.split('.')
    if parts[0] in _REFERENCE_TOPS or (parts[0] == 'src' and len(parts) > 1 and parts[1] in _REFERENCE_TOPS):
        cls = classes_by_name().get(name.split('.')[-1])
        if cls is not None:
            if log is not None:
                log.append((module, name, cls.__module__ + '.' + cls.__qualname__))
            return cls
    raise AttributeError('%s.%s' % (module, name))


def unresolved(path_or_bytes):
    """[(module, name)] a file needs and this build cannot supply, by path or by class name"""
    bad = []
    for mod, name in pickle_globals(path_or_bytes):
        try:
            resolve(mod, name)
        except AttributeError:
            bad.append((mod, name))
    return bad


class _Pickle:
    """what torch.load(pickle_module=...) needs: Unpickler / load / loads with the fallback find_class"""
    __name__ = 'aivc_amd.model_mngt.pickle_compat'

    def __init__(self, log):
        outer_log = log

        class Unpickler(pickle.Unpickler):
            def find_class(self, module, name):
                return resolve(module, name, outer_log)
        self.Unpickler = Unpickler
        self.Pickler = pickle.Pickler
        self.UnpicklingError = pickle.UnpicklingError
        self.HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL

    def load(self, f, **kw):
        return self.Unpickler(f, **kw).load()

    def loads(self, b, **kw):
        return self.Unpickler(io.BytesIO(b), **kw).load()


def resolver(log=None):
    """pickle_module for torch.load; `log` (a list) receives (module, name, resolved-to) for every class that was
    found by name instead of by path"""
    return _Pickle([] if log is None else log)
