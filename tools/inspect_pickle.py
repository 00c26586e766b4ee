#!/usr/bin/env python3
"""List the classes / functions a `.pt` (or any pickle) would import -- WITHOUT unpickling it -- and say how this
build resolves each: by the dotted path as written, by class name (the fallback of load_model for the `models`
package whose upstream layout is unknown), or not at all.

    python tools/inspect_pickle.py ../models/ms_ssim-2021cc-6/0_model.pt
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    from aivc_amd.model_mngt import pickle_compat
    rc = 0
    for mod, name in pickle_compat.pickle_globals(sys.argv[1]):
        log = []
        try:
            obj = pickle_compat.resolve(mod, name, log)
            how = ('by NAME -> ' + log[0][2]) if log else 'by path -> %s.%s' % (getattr(obj, '__module__', '?'), getattr(obj, '__qualname__', name))
        except AttributeError:
            how, rc = 'UNRESOLVED', 1
        print('%-60s %s' % (mod + '.' + name, how))
    return rc


if __name__ == '__main__':
    sys.exit(main())
