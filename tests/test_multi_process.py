"""world_size-2 gloo test of the unit sharding (aivc_amd/parallel.py): bytes gathered from two ranks
are identical to a single-process run.  The per-unit coder is the CPU oracle behind the FrameCodec
interface, so the test runs without a GPU; the sharding / gather / container logic under test is the
product's."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleUnitCodec:
    """FrameCodec look-alike (encode_video / assemble_video / decode_video with unit_filter)."""

    def __init__(self, spec):
        self.spec = spec

    def encode_video(self, frames, gop_name, idx_starting_frame=0, idx_end_frame=None, idx_rate=0., unit_filter=None):
        from oracle import codec as oc
        n = len(frames)
        unit = len(oc.gop_struct(gop_name))
        nb = math.ceil(n / unit)
        gops, data_dim = [None] * nb, None
        for u in range(nb):
            if unit_filter is not None and not unit_filter(u):
                continue
            chunk = [frames[min(u * unit + i, n - 1)] for i in range(unit)]
            blob, _ = oc.encode_video(self.spec, chunk, gop_name)
            gops[u] = oc.split_lp(blob, 18, 1)[0]
            v = [int.from_bytes(blob[i:i + 2], 'big') for i in range(0, 12, 2)]
            data_dim = {'x': (v[0], v[1]), 'y': (v[2], v[3]), 'z': (v[4], v[5])}
        return {'gops': gops, 'recs': None, 'data_dim': data_dim, 'nb_gop': nb,
                'idx_starting_frame': idx_starting_frame, 'idx_end_frame': idx_starting_frame + n - 1}

    @staticmethod
    def assemble_video(enc):
        from aivc_amd.codec import FrameCodec
        return FrameCodec.assemble_video(enc)

    def decode_video(self, blob, device=None, unit_filter=None):
        """FrameCodec.decode_video's contract: frames of units filtered out are None"""
        from aivc_amd.real_life import cat_binary_files as container
        from oracle import codec as oc
        data_dim, first, last, gops = container.unpack_video(blob)
        frames = []
        for u, g in enumerate(gops):
            n_unit = len(container.unpack_gop(g)[2])
            if unit_filter is not None and not unit_filter(u):
                frames += [None] * n_unit
                continue
            one = oc.video_header(data_dim, 1, 0, n_unit - 1) + oc.lp(g)
            frames += [{k: torch.from_numpy(f[k])[None] for k in 'yuv'} for f in oc.decode_video(self.spec, one)]
        return frames[:last - first + 1], data_dim, first, last


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from aivc_amd import parallel, synth
    from aivc_amd.models import arch
    from oracle import spec as ospec
    model = synth.make_model(arch.TINY_WIDTHS, seed=100 + rank)  # different weights per rank ...
    # something cached from the pre-broadcast weights (bench.py calibrates on the GPU before the broadcast)
    from aivc_amd.layers import _cache
    conv = model.codec_net.codec_net.g_a[0].layers[1]
    _cache.cached(conv, 'probe', (conv.weight,), lambda: conv.weight.detach().clone())
    parallel.broadcast_model(model)                            # ... until the broadcast
    probe = _cache.cached(conv, 'probe', (conv.weight,), lambda: conv.weight.detach().clone())
    assert torch.equal(probe, conv.weight), 'packed-parameter cache survived broadcast_model (stale weights)'
    frames = synth.synthetic_video(48, 32, 7, seed=2)
    codec = OracleUnitCodec(ospec.export_model(model))
    blob = parallel.encode_video_sharded(codec, frames, 'LDP_2')
    owners = [parallel.unit_owner(u, world) for u in range(3)]
    # decode: every rank holds the bitstream (broadcast from rank 0 for the test), decodes its units, rank 0 gets all
    n = torch.tensor([len(blob) if rank == 0 else 0])
    dist.broadcast(n, 0)
    buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).clone() if rank == 0 else torch.empty(int(n), dtype=torch.uint8)
    dist.broadcast(buf, 0)
    dec = parallel.decode_video_sharded(codec, buf.numpy().tobytes(), torch.device('cpu'))
    dec = None if dec is None else [{k: f[k][0].numpy().copy() for k in 'yuv'} for f in dec]
    if rank == 0:
        q.put((blob, owners, float(sum(p.double().sum() for p in model.parameters())), dec))
    else:
        q.put((None, owners, float(sum(p.double().sum() for p in model.parameters())), dec))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(oracle):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blobs = [r[0] for r in res if r[0] is not None]
    assert len(blobs) == 1
    assert res[0][1] == [0, 1, 0]
    assert abs(res[0][2] - res[1][2]) < 1e-9  # weights identical after the broadcast
    # single process reference with rank 0's weights
    from aivc_amd import synth
    from aivc_amd.models import arch
    from oracle import codec as oc
    from oracle import spec as ospec
    model = synth.make_model(arch.TINY_WIDTHS, seed=100)
    ref, ref_recs = oc.encode_video(ospec.export_model(model), synth.synthetic_video(48, 32, 7, seed=2), 'LDP_2')
    assert blobs[0] == ref
    decs = [r[3] for r in res if r[3] is not None]
    assert len(decs) == 1 and len(decs[0]) == 7  # rank 0 only; frames of both ranks' units, padding removed
    for d, r in zip(decs[0], ref_recs):
        assert all(np.array_equal(d[k], r[k]) for k in 'yuv')


class OracleFrameCodec:
    """encode_batch / max_batch look-alike of aivc_amd.codec.FrameCodec on the CPU oracle, returning
    ready frame bytes through a one-element 'sections' wrapper understood by finalize_frames' stub."""
    max_batch = 4

    def __init__(self, spec):
        self.spec = spec

    def encode_batch(self, cur, prev, nxt, frame_type, idx_rate=0.):
        from oracle import codec as oc
        recs, secs, dd = [], [], None

        def np_planes(p):
            return None if p is None else {k: p[k][0].numpy() for k in 'yuv'}
        for c, p, n in zip(cur, prev, nxt):
            fb, rec, dd = oc.encode_frame(self.spec, np_planes(c), np_planes(p), np_planes(n), frame_type, idx_rate)
            recs.append({k: torch.from_numpy(rec[k]).unsqueeze(0) for k in 'yuv'})
            secs.append(fb)
        return {'sections': secs, 'rec': recs, 'data_dim': dict(dd, x_uv=None)}

    def encode_units(self, units, gop_name, idx_rate=0., shard=None):
        from aivc_amd import parallel
        blobs, dd = parallel.encode_units_level_sharded(self, units, gop_name, idx_rate, shard=shard)
        return blobs, None, dd

    def decode_units(self, gop_blobs, data_dim, device=None, shard=None):
        from aivc_amd import parallel
        return parallel.decode_units_level_sharded(self, gop_blobs, data_dim, device or torch.device('cpu'), shard=shard)

    def decode_batch(self, frames_bytes, prev, nxt, frame_type, data_dim, idx_rate=0., device=None):
        from oracle import codec as oc

        def np_planes(p):
            return None if p is None else {k: p[k][0].numpy() for k in 'yuv'}
        out = []
        for fb, p, n in zip(frames_bytes, prev, nxt):
            rec = oc.decode_frame(self.spec, fb, np_planes(p), np_planes(n), frame_type, data_dim, idx_rate)
            out.append({k: torch.from_numpy(np.ascontiguousarray(rec[k])).unsqueeze(0) for k in 'yuv'})
        return out


def _worker_levels(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import aivc_amd.real_life.bitstream as bs
    bs.finalize_frames = lambda secs: list(secs)  # the oracle codec already returns frame bytes
    from aivc_amd import parallel, synth
    from aivc_amd.models import arch
    from oracle import spec as ospec
    model = synth.make_model(arch.TINY_WIDTHS, seed=100)
    frames = synth.synthetic_video(48, 32, 10, seed=2)
    units = [[{k: torch.from_numpy(f[k]).unsqueeze(0) for k in 'yuv'} for f in frames[u * 5:u * 5 + 5]] for u in range(2)]
    blobs, dd = parallel.encode_units_level_sharded(OracleFrameCodec(ospec.export_model(model)), units, '1_GOP_4')
    q.put((rank, blobs, dd['x'], dd['y'], dd['z']))
    dist.destroy_process_group()


def test_temporal_level_sharding_matches_single_process(oracle):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_levels, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]  # every rank ends with the same GOP records
    from aivc_amd import synth
    from aivc_amd.models import arch
    from oracle import codec as oc
    from oracle import spec as ospec
    model = synth.make_model(arch.TINY_WIDTHS, seed=100)
    ref, _ = oc.encode_video(ospec.export_model(model), synth.synthetic_video(48, 32, 10, seed=2), '1_GOP_4')
    assert oc.split_lp(ref, 18, 2) == res[0][1]
    assert (res[0][2], res[0][3], res[0][4]) == ((32, 48), tuple(int.from_bytes(ref[i:i + 2], 'big') for i in (4, 6)),
                                                  tuple(int.from_bytes(ref[i:i + 2], 'big') for i in (8, 10)))


def _worker_levels_decode(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from aivc_amd import parallel, synth
    from aivc_amd.models import arch
    from oracle import codec as oc
    from oracle import spec as ospec
    model = synth.make_model(arch.TINY_WIDTHS, seed=100)
    spec = ospec.export_model(model)
    blob, _ = oc.encode_video(spec, synth.synthetic_video(48, 32, 10, seed=2), '1_GOP_4')
    gops = oc.split_lp(blob, 18, 2)
    dd = {'x': (32, 48), 'y': tuple(int.from_bytes(blob[i:i + 2], 'big') for i in (4, 6)),
          'z': tuple(int.from_bytes(blob[i:i + 2], 'big') for i in (8, 10))}
    recs = parallel.decode_units_level_sharded(OracleFrameCodec(spec), gops, dd, device=torch.device('cpu'))
    digest = [bytes(torch.cat([fr[k].reshape(-1) for k in 'yuv']).numpy()) for unit in recs for fr in unit]
    q.put((rank, digest))
    dist.destroy_process_group()


def test_temporal_level_sharded_decode_matches_single_process(oracle):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_levels_decode, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]  # every rank ends with the same frames
    from aivc_amd import synth
    from aivc_amd.models import arch
    from oracle import codec as oc
    from oracle import spec as ospec
    spec = ospec.export_model(synth.make_model(arch.TINY_WIDTHS, seed=100))
    blob, _ = oc.encode_video(spec, synth.synthetic_video(48, 32, 10, seed=2), '1_GOP_4')
    frames = oc.decode_video(spec, blob)
    want = [bytes(np.concatenate([np.asarray(fr[k]).reshape(-1) for k in 'yuv'])) for fr in frames]
    assert res[0][1] == want


# ---- one clip over 4 ranks: 2 unit groups x 2 ranks of level sharding (the layout of BASELINE configs[3] on 8 GPUs)
def _worker_clip(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import aivc_amd.real_life.bitstream as bs
    bs.finalize_frames = lambda secs: list(secs)  # the oracle codec already returns frame bytes
    from aivc_amd import parallel, synth
    from aivc_amd.models import arch
    from oracle import spec as ospec
    model = synth.make_model(arch.TINY_WIDTHS, seed=100)
    frames = synth.synthetic_video(48, 32, 10, seed=2)
    units = [[{k: torch.from_numpy(f[k]).unsqueeze(0) for k in 'yuv'} for f in frames[u * 5:u * 5 + 5]] for u in range(2)]
    codec = OracleFrameCodec(ospec.export_model(model))
    shard = parallel.ClipShard(len(units), torch.device('cpu'))
    blobs, dd = parallel.encode_clip(codec, units, '1_GOP_4', shard=shard)
    recs = parallel.decode_clip(codec, blobs, dd, torch.device('cpu'), shard=shard)
    digest = {u: [bytes(torch.cat([fr[k].reshape(-1) for k in 'yuv']).numpy()) for fr in frs] for u, frs in recs.items()}
    q.put((rank, blobs, (dd['x'], dd['y'], dd['z']), (shard.G, shard.R, shard.group_id, shard.local, shard.units), digest))
    dist.destroy_process_group()


def test_clip_over_unit_groups_and_levels_matches_single_process(oracle):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_clip, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[3] for r in res] == [(2, 2, 0, 0, [0]), (2, 2, 0, 1, [0]), (2, 2, 1, 0, [1]), (2, 2, 1, 1, [1])]
    assert all(r[1] == res[0][1] and r[2] == res[0][2] for r in res)  # every rank holds the whole bitstream
    from aivc_amd import synth
    from aivc_amd.models import arch
    from oracle import codec as oc
    from oracle import spec as ospec
    spec = ospec.export_model(synth.make_model(arch.TINY_WIDTHS, seed=100))
    ref, _ = oc.encode_video(spec, synth.synthetic_video(48, 32, 10, seed=2), '1_GOP_4')
    assert oc.split_lp(ref, 18, 2) == res[0][1]
    frames = oc.decode_video(spec, ref)
    want = [bytes(np.concatenate([np.asarray(fr[k]).reshape(-1) for k in 'yuv'])) for fr in frames]
    for r in res:
        (u, got), = r[4].items()  # each rank holds the frames of its group's unit
        assert got == want[5 * u:5 * u + 5]


def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


# ---- row bands (aivc_amd/bands.py) over REAL processes: the band engine with its torch.distributed transport
# (DistComm: batch_isend_irecv between neighbours + all_gather inside a sub-group, host tensors under gloo) and the
# oracle's conv as the per-slab kernel -- the CPU twin of tests/test_gpu_multi_process.py's banded cases
def _band_worker(rank, world, port, q, h_y, seed):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), AIVC_DIST_BACKEND='gloo')
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from aivc_amd import parallel
    from aivc_amd.bands import BandCtx, DistComm
    from oracle import oracle as orc
    import band_chain
    parallel.init_process_group()
    orc.lib()
    pg = dist.new_group(list(range(world)))  # (the product builds its bands on a sub-group: ClipShard.bands())
    ctx = BandCtx(DistComm(pg, list(range(world)), rank), torch.device('cpu'))
    x, wts, H = band_chain.make_case(h_y, seed)
    full, (v0, v1), rows = band_chain.banded(orc, ctx, x, wts, h_y, H)
    q.put((rank, full, v0, v1, rows, dict(ctx.comm.stats)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,h_y,seed', [(2, 5, 11), (3, 7, 12), (4, 3, 13)])
def test_row_bands_over_gloo_processes(world, h_y, seed, oracle):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import band_chain
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, q, h_y, seed)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, wts, H = band_chain.make_case(h_y, seed)
    t2, want = band_chain.whole(oracle, x, wts)
    covered = 0
    for rank, full, v0, v1, rows, stats in res:
        np.testing.assert_array_equal(full, t2)
        np.testing.assert_array_equal(rows, want[:, v0:v1])
        covered += v1 - v0
    assert covered == want.shape[1]
    assert max(r[5]['bytes_sent'] for r in res) > 0 and all(r[5]['gathers'] == 1 for r in res)
