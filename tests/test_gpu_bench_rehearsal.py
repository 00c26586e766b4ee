"""Rehearsal of the driver's SCALE command on the one GPU of the test box: `python bench.py --gpus N` end to end --
its own re-launch under torch.distributed.run, one process per rank, ClipShard (unit groups x temporal-layer
sharding), the warm-up byte check against the single-rank encode, the strong line with its weak-scaling object,
the final barrier and a clean exit of every rank -- with every rank on cuda:0 (AIVC_BENCH_SINGLE_DEVICE=1) and gloo
in RCCL's place (AIVC_DIST_BACKEND=gloo; RCCL refuses two ranks per device).  The first real 8-GPU run must not also
be the first run of that script's N > 1 branches.  Not a measurement: the line says so (`invalid`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra=(), timeout=900):
    env = dict(os.environ, AIVC_BENCH_SINGLE_DEVICE='1', AIVC_DIST_BACKEND='gloo', AIVC_NO_QUALITY='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONDONTWRITEBYTECODE='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '1',
           '--width', '416', '--height', '240', '--frames', '64', '--gop', '1_GOP_16', '--no-cpu-baseline',
           '--no-roofline', '--no-high-rate', '--no-lean-encoder', '--no-precision-mode'] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, 'bench.py --gpus %d exited %d\n%s' % (n, r.returncode, r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, 'stdout must carry exactly one line, got %d:\n%s' % (len(lines), r.stdout[-2000:])
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize('n', [2, 4])
def test_bench_multi_rank_flow_end_to_end(n, cuda):
    """416x240, 64 frames under `1_GOP_16` = 4 intra-period units of 17 frames (the clip of configs[3] in small): on 2
    ranks two groups ... on 4 ranks four groups of one rank; strong scaling is the headline, weak beside it."""
    out, err = _run(n)
    assert out['n_gpus'] == n and out['steps'] == 1 and out['warmup'] == 1
    assert out['scaling'] == 'strong'
    assert out['bytes_equal_single_rank'] is True and out['closed_loop_ok'] is True
    assert out['stream_errors_rank0'] == 0
    assert out['value'] > 0 and out['unit'] == 'frames/s' and out['higher_is_better'] is True
    assert abs(out['value'] - 64 / (out['ms_per_step'] * 1e-3)) < 0.02 * out['value']
    assert out['config']['units_per_step'] == 4 and out['config']['coded_frames_per_step'] == 68
    weak = out['weak_scaling']
    assert weak['scaling'] == 'weak' and weak['value'] > 0
    assert abs(weak['value'] - n * 64 / (weak['ms_per_step'] * 1e-3)) < 0.02 * weak['value']
    assert 'gloo' in out['invalid']  # a validation run says so in its line
    assert 'Traceback' not in err


def test_bench_one_unit_over_four_ranks_level_sharding_and_bands(cuda):
    """ONE intra-period unit on 4 ranks (what configs[4] asks of 8 GPUs): one group, the frames of a dependency level
    dealt over its ranks, the levels narrower than the group in row bands (4 ranks: the automatic rule) after the
    warm-up clip came out byte-identical to the single-rank encode"""
    out, err = _run(4, ['--frames', '16', '--gop', '1_GOP_16', '--contract', 'fp32'])  # (row bands: version 1 of the contract only)
    assert out['n_gpus'] == 4 and out['scaling'] == 'strong' and out['arithmetic_contract'] == 'fp32'
    assert out['bytes_equal_single_rank'] is True and out['closed_loop_ok'] is True
    assert out['config']['units_per_step'] == 1
    assert out['row_bands'] is not None and out['row_bands'].startswith('on')
    assert 'x4' in out['config']['parallelism']


def test_bench_one_unit_over_four_ranks_in_the_default_contract(cuda):
    """the same in the default (version 2) contract: level sharding only, row bands reported off, bytes still those of one rank"""
    out, err = _run(4, ['--frames', '16', '--gop', '1_GOP_16', '--contract', 'fp32w'])
    assert out['n_gpus'] == 4 and out['scaling'] == 'strong' and out['arithmetic_contract'] == 'fp32w'
    assert out['bytes_equal_single_rank'] is True and out['closed_loop_ok'] is True
    assert out['row_bands'] is not None and out['row_bands'].startswith('off')


def test_bench_weak_scaling_flag(cuda):
    out, _ = _run(2, ['--scaling', 'weak'])
    assert out['scaling'] == 'weak' and out['n_gpus'] == 2 and out['closed_loop_ok'] is True
    assert abs(out['value'] - 2 * 64 / (out['ms_per_step'] * 1e-3)) < 0.02 * out['value']
    assert 'weak_scaling' not in out


def test_bench_two_ranks_at_a_size_version_2_covers(cuda):
    """1280x720: the 3x3 layers at 1/4 resolution (320 x 180) and both 5x5 forms run their Winograd kernels here (416x240 above is
    below the contract's size rules) -- the sharded bitstream must still be the single rank's: a Winograd chain depends on the image,
    never on the batch it is launched in"""
    out, err = _run(2, ['--width', '1280', '--height', '720', '--frames', '18', '--gop', '1_GOP_8', '--contract', 'fp32w'])
    assert out['n_gpus'] == 2 and out['arithmetic_contract'] == 'fp32w'
    assert out['bytes_equal_single_rank'] is True and out['closed_loop_ok'] is True
    assert out['stream_errors_rank0'] == 0
