#!/usr/bin/env python3
"""First contact with RCCL on a multi-GPU node, one named step at a time (round-4 review item 8).

The builder's boxes have ONE GPU: every N > 1 path of aivc_amd/parallel.py and aivc_amd/bands.py has been exercised over
gloo (real processes) and with RCCL at world size 1 only.  Run this BEFORE `bench.py --gpus N` on the 8-GPU node:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
        tools/rccl_preflight.py [--timeout 30] [--codec]

Every step runs under its own deadline (a watchdog thread: when a collective does not return in --timeout seconds the
step's name and the rank are printed and the process exits with status 3, instead of hanging the launcher), and rank 0
prints ONE line per step:  `[preflight] <step>: ok 12.3 ms` / `FAILED: <what>`, then a JSON verdict.  Steps:

  init            init_process_group('nccl', device_id=cuda:LOCAL_RANK), finite timeout
  all_reduce      a 4-byte all_reduce on the world (the first communicator use)
  broadcast       one flat 64 MB fp32 broadcast (what broadcast_model sends: the weights in one tensor)
  new_group GxR   dist.new_group for every ClipShard layout of this world size (4x2 and 1x8 at N = 8, units = 4 / 1),
                  in the same order on every rank, then an all_reduce inside this rank's group
  all_gather      all_gather_into_tensor of one 3.1 MB uint8 frame per rank inside the group (exchange_frames)
  ring            batch_isend_irecv to both neighbours of the group (the halo exchange of the row bands, bands.DistComm)
  codec (--codec) a 2-unit tiny-model clip through parallel.encode_clip / decode_clip on every layout, banded levels
                  forced on: bytes == this rank's own single-process encode, decoder == encoder reconstruction

Nothing here is a measurement; it names the step that fails."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)


class Deadline:
    """`with Deadline(name, seconds):` -- a step that does not finish in time ends the process with its name"""
    current = None

    def __init__(self, name, seconds, rank):
        self.name, self.seconds, self.rank = name, seconds, rank

    def __enter__(self):
        self.t0 = time.time()
        self.done = threading.Event()

        def watch():
            if not self.done.wait(self.seconds):
                sys.stderr.write('[preflight] rank %d: step "%s" did not return within %.0f s -- giving up\n'
                                 % (self.rank, self.name, self.seconds))
                sys.stderr.flush()
                os._exit(3)
        threading.Thread(target=watch, daemon=True).start()
        return self

    def __exit__(self, *exc):
        self.done.set()
        self.ms = (time.time() - self.t0) * 1e3
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--timeout', type=float, default=30.0, help='seconds per step')
    ap.add_argument('--codec', action='store_true', help='also run a tiny clip through the sharded codec paths')
    ap.add_argument('--backend', default=os.environ.get('AIVC_DIST_BACKEND', 'nccl'))
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = 0 if os.environ.get('AIVC_SINGLE_DEVICE') else int(os.environ.get('LOCAL_RANK', '0'))
    results = []

    def report(step, ok, ms, detail=''):
        results.append({'step': step, 'ok': bool(ok), 'ms': round(ms, 1), 'detail': detail})
        if rank == 0:
            print('[preflight] %s: %s %.1f ms %s' % (step, 'ok' if ok else 'FAILED', ms, detail), flush=True)

    def step(name, fn):
        try:
            with Deadline(name, a.timeout, rank) as d:
                detail = fn() or ''
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            report(name, True, d.ms, detail)
            return True
        except Exception as e:  # noqa: BLE001 -- the point is to name the step
            report(name, False, 0.0, '%s: %s' % (type(e).__name__, e))
            return False

    if not torch.cuda.is_available():
        print('[preflight] no GPU visible')
        return 2
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from aivc_amd import parallel
    cdev = dev if a.backend == 'nccl' else torch.device('cpu')

    def do_init():
        parallel.init_process_group(dev, a.backend, timeout_s=max(a.timeout * 4, 60))
        return 'backend %s, world %d, %s' % (dist.get_backend(), dist.get_world_size(), torch.cuda.get_device_name(local))
    if not step('init', do_init):
        return 1

    def do_allreduce():
        t = torch.ones(1, device=cdev)
        dist.all_reduce(t)
        assert int(t.item()) == world, t
    step('all_reduce', do_allreduce)

    def do_broadcast():
        t = torch.full((16 << 20,), float(rank), device=cdev)
        dist.broadcast(t, 0)
        assert float(t[0]) == 0.0 and float(t[-1]) == 0.0
        return '64 MB fp32'
    step('broadcast', do_broadcast)

    layouts = sorted({min(world, u) for u in (4, 1)}, reverse=True)  # unit groups G for 4-unit and 1-unit clips
    shards = {}
    for g_units in layouts:
        n_units = 4 if g_units > 1 or world == 1 else 1

        def do_groups(n_units=n_units):
            sh = parallel.ClipShard(n_units, dev)
            shards[n_units] = sh
            if sh.pg is not None:
                t = torch.ones(1, device=cdev)
                dist.all_reduce(t, group=sh.pg)
                assert int(t.item()) == sh.R
            return 'G %d x R %d' % (sh.G, sh.R)
        step('new_group units=%d' % n_units, do_groups)
    for n_units, sh in shards.items():
        if sh.pg is None or sh.R <= 1:
            continue

        def do_gather(sh=sh):
            fsz = 1920 * 1080 * 3 // 2
            mine = torch.full((fsz,), sh.local, dtype=torch.uint8, device=cdev)
            out = torch.empty((sh.R * fsz,), dtype=torch.uint8, device=cdev)
            dist.all_gather_into_tensor(out, mine, group=sh.pg)
            got = out.view(sh.R, fsz)[:, 0].cpu().tolist()
            assert got == list(range(sh.R)), got
            return '%d x %.1f MB uint8' % (sh.R, fsz / 1e6)
        step('all_gather frame (R=%d)' % sh.R, do_gather)

        def do_ring(sh=sh):
            base = sh.group_id * sh.R
            up, down = base + (sh.local - 1) % sh.R, base + (sh.local + 1) % sh.R
            send = torch.full((1 << 18,), float(sh.local), device=cdev)
            r_up, r_down = torch.empty_like(send), torch.empty_like(send)
            ops = [dist.P2POp(dist.isend, send, down, group=sh.pg), dist.P2POp(dist.isend, send, up, group=sh.pg),
                   dist.P2POp(dist.irecv, r_up, up, group=sh.pg), dist.P2POp(dist.irecv, r_down, down, group=sh.pg)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            assert float(r_up[0]) == (sh.local - 1) % sh.R and float(r_down[0]) == (sh.local + 1) % sh.R
            return '1 MB to both neighbours'
        step('ring isend/irecv (R=%d)' % sh.R, do_ring)

    if a.codec:
        from aivc_amd import synth
        from aivc_amd.codec import FrameCodec
        from aivc_amd.models import arch
        model = synth.make_model(arch.TINY_WIDTHS, seed=7, device=dev)
        parallel.broadcast_model(model)
        fc = FrameCodec(model)
        for n_units, sh in shards.items():
            def do_codec(n_units=n_units, sh=sh):
                os.environ['AIVC_BAND_LEVELS'] = '1'  # the row-band transport too
                frames = synth.to_device_frames(synth.synthetic_video(192, 128, 9 * n_units, seed=3), dev)
                units = [frames[9 * u:9 * (u + 1)] for u in range(n_units)]
                with torch.no_grad():
                    blobs, dd = parallel.encode_clip(fc, units, '1_GOP_8', shard=sh)
                    ref, recs, _ = fc.encode_units(units, '1_GOP_8')
                    dec = parallel.decode_clip(fc, blobs, dd, dev, shard=sh)
                assert blobs == ref, 'sharded bytes differ from the single-process bytes'
                for u, frs in dec.items():
                    assert all(torch.equal(d[k], e[k]) for d, e in zip(frs, recs[u]) for k in 'yuv'), 'decoder != encoder'
                return '%d unit(s), G %d x R %d, banded levels on' % (n_units, sh.G, sh.R)
            step('codec units=%d' % n_units, do_codec)

    ok = all(r['ok'] for r in results)
    flag = torch.tensor([0 if ok else 1], device=cdev)
    with Deadline('final all_reduce', a.timeout, rank):
        dist.all_reduce(flag)
    if rank == 0:
        print(json.dumps({'preflight': 'ok' if int(flag.item()) == 0 else 'FAILED', 'world': world, 'backend': a.backend,
                          'ranks_with_failures': int(flag.item()), 'steps': results}))
    dist.destroy_process_group()
    return 0 if int(flag.item()) == 0 else 1


if __name__ == '__main__':
    sys.exit(main())
