"""Strong-scaling path (aivc_amd/parallel.py: ClipShard, encode_clip / decode_clip) with the HIP codec and REAL
process boundaries on the one GPU of the test box: two processes share cuda:0 for the kernels and talk over gloo
(host tensors) -- RCCL refuses two ranks on one device.  What this covers that the CPU gloo tests cannot: the
level-sharded FrameCodec.encode_units / decode_units (side streams, deferred flags, per-level exchange) produce
the bytes and frames of a single process."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _setup(n_units, gop, w=80, h=48, default_widths=False):
    from aivc_amd import synth
    from aivc_amd.func_util.GOP_structure import generate_gop_struct
    from aivc_amd.models import arch
    dev = torch.device('cuda:0')
    if default_widths:
        model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=dev)
        synth.calibrate_operating_point(model, dev)
    else:
        model = synth.make_model(arch.TINY_WIDTHS, seed=77, device=dev)
    unit = len(generate_gop_struct(gop))
    frames = synth.to_device_frames(synth.synthetic_video(w, h, unit * n_units, seed=4), dev)
    units = [frames[u * unit:(u + 1) * unit] for u in range(n_units)]
    return model, units, dev


def _worker(rank, world, port, q, n_units, gop, w, h, default_widths, contract=None):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    if contract:  # row bands exist under version 1 of the arithmetic contract only (FrameCodec._banded)
        os.environ['AIVC_CONTRACT'] = contract
    if w * h < 6000000:
        os.environ['AIVC_BAND_LEVELS'] = '1'  # small frames: the automatic rule would not band 2 ranks (FrameCodec._banded)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), AIVC_DIST_BACKEND='gloo')
    from aivc_amd import parallel
    parallel.init_process_group()  # (the product's start-up: timeout + watchdog)
    model, units, dev = _setup(n_units, gop, w, h, default_widths)
    parallel.broadcast_model(model)
    fc = model.frame_codec()
    shard = parallel.ClipShard(n_units, dev)
    with torch.no_grad():
        blobs, dd = parallel.encode_clip(fc, units, gop, shard=shard)
        recs = parallel.decode_clip(fc, blobs, dd, dev, shard=shard)
    torch.cuda.synchronize()
    digest = {u: [bytes(torch.cat([fr[k].reshape(-1) for k in 'yuv']).cpu().numpy()) for fr in frs] for u, frs in recs.items()}
    bands = getattr(shard, '_bands', None)
    q.put((rank, blobs, (shard.G, shard.R), digest, None if bands is None else (bands.launches, dict(bands.comm.stats))))
    dist.destroy_process_group()


# (1, '1_GOP_8') on 2 ranks and (1, '1_GOP_4') on 4: ONE unit, so its 1- (and 2-) frame levels are narrower than the group and
# are coded in ROW BANDS with halo exchange (aivc_amd/bands.py) -- odd frame size, more ranks than some levels have frames,
# and BASELINE configs[4]'s frame size with the default-width model (its 1_GOP_2: three single-frame levels)
@pytest.mark.parametrize('n_units,gop,world,layout,w,h,default_widths,banded', [
    (1, '1_GOP_8', 2, (1, 2), 80, 48, False, True), (2, '1_GOP_4', 2, (2, 1), 80, 48, False, False),
    (1, '2_GOP_4', 2, (1, 2), 80, 48, False, True), (1, '1_GOP_8', 2, (1, 2), 83, 57, False, True),
    (1, '1_GOP_4', 4, (1, 4), 96, 80, False, True), (1, '1_GOP_2', 2, (1, 2), 3840, 2160, True, True)])
def test_processes_on_one_gpu_match_single_process(n_units, gop, world, layout, w, h, default_widths, banded, cuda):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_units, gop, w, h, default_widths, 'fp32' if banded else None)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[2] == layout for r in res)
    assert all(r[1] == res[0][1] for r in res)
    # row bands were (not) used, and what travelled were halo rows, not activations
    for r in res:
        assert (r[4] is not None and r[4][0] > 0) == banded
    from aivc_amd import ops
    model, units, dev = _setup(n_units, gop, w, h, default_widths)
    fc = model.frame_codec()
    prev = ops.set_precision('fp32') if banded else None  # the single-process reference in the workers' contract
    try:
        with torch.no_grad():
            ref_blobs, ref_recs, dd = fc.encode_units(units, gop)
            ref_dec = fc.decode_units(ref_blobs, dd, dev)
    finally:
        if prev:
            ops.set_precision(prev)
    assert res[0][1] == ref_blobs
    for r in res:
        for u, got in r[3].items():
            want = [bytes(torch.cat([fr[k].reshape(-1) for k in 'yuv']).cpu().numpy()) for fr in ref_dec[u]]
            assert got == want
            enc = [bytes(torch.cat([fr[k].reshape(-1) for k in 'yuv']).cpu().numpy()) for fr in ref_recs[u]]
            assert got == enc  # decoder == encoder reconstruction


def test_single_rank_clip_shard_is_the_plain_path(cuda):
    from aivc_amd import parallel
    model, units, dev = _setup(2, 'LDP_2')
    fc = model.frame_codec()
    with torch.no_grad():
        blobs, dd = parallel.encode_clip(fc, units, 'LDP_2')
        ref_blobs, _, ref_dd = fc.encode_units(units, 'LDP_2')
        recs = parallel.decode_clip(fc, blobs, dd, dev)
    assert blobs == ref_blobs and dd == ref_dd and sorted(recs) == [0, 1]


def test_exchange_frames_device_side_plumbing(cuda, monkeypatch):
    """ClipShard.exchange_frames with the tensors an RCCL run hands it (uint8 CUDA planes, CUDA send / receive buffers,
    asynchronous collective): the box has one GPU and RCCL refuses two ranks per device, so the collective itself is
    replaced by a stand-in that delivers what rank 1 of a two-rank group would send -- packing, slot order and the
    views handed back are the product's."""
    import torch.distributed as dist
    from aivc_amd import parallel
    h, w = 34, 50
    hc, wc = 17, 25
    g = torch.Generator(device='cpu').manual_seed(5)

    def frame():
        return {'y': torch.randint(0, 256, (1, h, w), generator=g, dtype=torch.uint8).to(cuda),
                'u': torch.randint(0, 256, (1, hc, wc), generator=g, dtype=torch.uint8).to(cuda),
                'v': torch.randint(0, 256, (1, hc, wc), generator=g, dtype=torch.uint8).to(cuda)}
    items = [(0, 'frame_%d' % i) for i in range(5)]  # 5 frames over 2 ranks: rank 0 codes 0, 2, 4; rank 1 codes 1, 3
    frames = [frame() for _ in items]
    sh = parallel.ClipShard.__new__(parallel.ClipShard)
    sh.rank, sh.world, sh.G, sh.R, sh.local, sh.pg, sh.device, sh._bufs = 0, 2, 1, 2, 0, None, cuda, {}
    other = torch.cat([torch.cat([frames[j][k].reshape(-1) for k in 'yuv']) for j in (1, 3)])

    class Work:
        def wait(self):
            return True

    def fake_all_gather(recv, send, group=None, async_op=False):
        assert recv.is_cuda and send.is_cuda and recv.dtype == torch.uint8 and async_op
        per = send.numel()
        recv[:per] = send
        recv[per:per + other.numel()] = other
        return Work()
    monkeypatch.setattr(dist, 'all_gather_into_tensor', fake_all_gather)
    monkeypatch.setattr(dist, 'get_backend', lambda group=None: 'nccl')
    got = sh.exchange_frames(items, [frames[j] for j in (0, 2, 4)], h, w, cuda)
    assert len(got) == 5
    for a, b in zip(got, frames):
        assert all(torch.equal(a[k], b[k]) and a[k].shape == b[k].shape for k in 'yuv')
    # the persistent send buffer is reused by the next level
    got2 = sh.exchange_frames(items, [frames[j] for j in (0, 2, 4)], h, w, cuda)
    assert len(sh._bufs) == 1 and all(torch.equal(a[k], b[k]) for a, b in zip(got2, frames) for k in 'yuv')


def test_gather_bytes_device_side_plumbing(cuda, monkeypatch):
    """gather_bytes_all as an RCCL run drives it (int64 lengths and uint8 payload as CUDA tensors), the second rank's
    contribution supplied by a stand-in collective"""
    import torch.distributed as dist
    from aivc_amd import parallel
    keys = [0, 1, 2, 3, 'dd']
    mine = {0: b'unit zero', 2: b'\x00\x01\x02' * 50, 'dd': b'D' * 24}
    theirs = {1: b'', 3: b'unit three!'}
    calls = []

    def fake_all_gather(recv, send, group=None, async_op=False):
        assert recv.is_cuda and send.is_cuda
        n = send.numel()
        recv[:n] = send
        if send.dtype == torch.int64:
            recv[n:] = torch.tensor([len(theirs[k]) if k in theirs else -1 for k in keys], dtype=torch.int64, device=send.device)
        else:
            pay = b''.join(theirs[k] for k in keys if k in theirs)
            recv[n:n + len(pay)] = torch.frombuffer(bytearray(pay), dtype=torch.uint8).to(send.device)
        calls.append(send.dtype)
    monkeypatch.setattr(dist, 'all_gather_into_tensor', fake_all_gather)
    monkeypatch.setattr(dist, 'get_backend', lambda group=None: 'nccl')
    out = parallel.gather_bytes_all(mine, keys, None, 2, cuda)
    want = dict(mine)
    want.update(theirs)
    assert out == want and calls == [torch.int64, torch.uint8]


def test_cli_two_ranks_equals_one_process(cuda, tmp_path):
    """aivc.py under `torch.distributed.run` with two ranks (intra-period units sharded, rank 0 writes the files and
    evaluates): bitstream and decoded .yuv are byte-identical to the single-process run, and the four evaluate lines
    are printed once.  Both ranks sit on the box's one GPU over gloo (RCCL refuses two ranks per device)."""
    import subprocess
    import sys
    import numpy as np
    from aivc_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w, h, n = 96, 64, 10  # RA 2/4: three intra-period units of 5, 4 and (padded) 1 frames
    raw = tmp_path / ('clip_%dx%d_30_420.yuv' % (w, h))
    with open(raw, 'wb') as f:
        for fr in synth.synthetic_video(w, h, n):
            for k in 'yuv':
                f.write(fr[k].tobytes())
    common = ['-i', str(raw), '--coding_config', 'RA', '--gop_size', '2', '--intra_period', '4', '--start_frame', '0',
              '--end_frame', str(n - 1)]
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    one = subprocess.run([sys.executable, '-m', 'aivc_amd.aivc'] + common + ['--bitstream_out', str(tmp_path / 'a.bin'), '-o', str(tmp_path / 'a.yuv')],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    port = _free_port()
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), '-m', 'aivc_amd.aivc'] + common + ['--bitstream_out', str(tmp_path / 'b.bin'), '-o', str(tmp_path / 'b.yuv')],
                         cwd=str(tmp_path), env=dict(env, AIVC_DIST_BACKEND='gloo', AIVC_SINGLE_DEVICE='1'), capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    assert (tmp_path / 'a.bin').read_bytes() == (tmp_path / 'b.bin').read_bytes()
    assert np.array_equal(np.fromfile(tmp_path / 'a.yuv', np.uint8), np.fromfile(tmp_path / 'b.yuv', np.uint8))
    pick = lambda out: [l for l in out.splitlines() if l.startswith(('PSNR    [dB]', 'MS-SSIM     ', 'MS-SSIM [dB]', 'Size [bytes]'))]
    assert len(pick(two.stdout)) == 4 and pick(two.stdout) == pick(one.stdout)
