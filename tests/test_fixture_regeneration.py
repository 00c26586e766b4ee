"""Fixture hygiene: both generators are deterministic -- re-running them in the build container (where the
reference is mounted) reproduces every committed fixture bit for bit.  Skipped wherever /root/reference is
absent (the GPU box), since the generators import the reference's modules."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/src'), reason='reference not mounted here')


@pytest.mark.parametrize('script,prefixes', [('gen_golden.py', None), ('gen_golden_decoder.py', ('decoder_',))])
def test_generators_reproduce_committed_fixtures(script, prefixes, tmp_path):
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', script), '--out', str(tmp_path)], check=True,
                   stdout=subprocess.DEVNULL, timeout=900)
    made = sorted(os.listdir(tmp_path))
    assert made
    for name in made:
        a, b = np.load(tmp_path / name), np.load(os.path.join(GOLDEN, name))
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            assert a[k].tobytes() == b[k].tobytes(), (name, k)
    if prefixes is None:  # every committed fixture that is not made by another generator is covered
        others = ('decoder_', 'metrics')
        committed = {f for f in os.listdir(GOLDEN) if f.endswith('.npz') and not f.startswith(others)}
        assert committed == set(made)
