#!/usr/bin/env python3
"""bench.py -- encode+decode throughput of the AIVC hot path on MI355X.

One "step" = encode + decode of one intra-period unit (33 frames: I, P, 31 hierarchical B) of
synthetic 1920x1080 8-bit YUV 4:2:0 video under random-access coding `1_GOP_32` (BASELINE.json
configs[3], the configuration the metric is quoted on; it fits one GPU).  Units are independent
(own I frame), so with N GPUs each rank codes its own units (weak scaling, no data-path collective;
one broadcast of the weights at start).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (see the driver contract) with two extra objects:
  roofline      dominant kernel (fp32 MFMA implicit-GEMM conv): algorithmic FLOPs per launch / average
                launch duration, measured with HIP events on the launch stream during an extra,
                untimed, instrumented step.
  cpu_baseline  the CPU oracle (a port, oracle/) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
_TILES = {0: '128x128', 1: '64x64', 2: '256x64', 3: '128x32', 4: '256x128', 5: '64x128'}
_MODES = {0: 'conv', 1: 'tconv', 2: 'gdn'}


def variant_name(v):
    """aivc_conv2d_variant code -> readable kernel name"""
    if v == 0:
        return 'conv_direct_kernel'
    if v == 1:
        return 'thin_tconv_kernel'
    if v == 2:
        return 'thin_mfma_kernel'
    c = v - 100
    fused = c >= 50
    c -= 50 if fused else 0
    return 'conv_mfma<%s,%s%s>' % (_MODES.get(c // 10, '?'), _TILES.get(c % 10, '?'), '+gdn' if fused else '')


class _Names(dict):
    def get(self, k, default=None):
        return variant_name(k)

    def __contains__(self, k):
        return True


VARIANT_NAMES = _Names()


def gpu_synthetic_unit(width, height, n_frames, t0, device, seed):
    """Same moving pattern as aivc_amd.synth.synthetic_video, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    hc, wc = (height + 1) // 2, (width + 1) // 2
    xs = torch.arange(width, device=device).float()[None, :]
    ys = torch.arange(height, device=device).float()[:, None]
    xc = torch.arange(wc, device=device).float()[None, :]
    yc = torch.arange(hc, device=device).float()[:, None]
    two_pi = 6.283185307179586
    out = []
    for t in range(t0, t0 + n_frames):
        y = 128 + 64 * torch.sin(two_pi * (xs + 3 * t) / 97) + 48 * torch.cos(two_pi * (ys - 2 * t) / 61)
        u = 128 + 40 * torch.sin(two_pi * (xc + 1.5 * t) / 53) * torch.cos(two_pi * yc / 47)
        v = 128 + 40 * torch.cos(two_pi * (yc - t) / 41) * torch.sin(two_pi * xc / 59)
        f = {}
        for k, a in (('y', y), ('u', u), ('v', v)):
            a = a + 4 * torch.randn(a.shape, device=device, generator=g)
            f[k] = a.round().clamp(0, 255).to(torch.uint8).unsqueeze(0).contiguous()
        out.append(f)
    return out


def cpu_baseline(width, height, model):
    """Oracle (CPU port) encode+decode of an I + P pair; falls back to a quarter-size frame when a
    probe says the full-size sample would exceed ~40 s."""
    import numpy as np
    from aivc_amd import synth
    from oracle import codec as ocodec
    from oracle import oracle as orc
    from oracle import spec as ospec
    orc.lib()
    cores = os.cpu_count() or 1
    spec = ospec.export_model(model)
    # probe: one 3x3 128->128 conv on a 135x240 map (9.6 GFLOP)
    x = np.random.default_rng(0).standard_normal((1, 135, 240, 128), dtype=np.float32)
    w = np.random.default_rng(1).standard_normal((128, 3, 3, 128), dtype=np.float32) * 0.03
    t = time.time()
    orc.conv2d(x, w, None, pad=1)
    gflops = 9.56 / max(time.time() - t, 1e-6)
    est_full = 3200.0 / gflops  # ~3.2 TFLOP for I + P at 1080p with the default widths
    scale = 1
    w_s, h_s = width, height
    if est_full > 40.0:
        scale, w_s, h_s = 4, width // 2, height // 2
    frames = synth.synthetic_video(w_s, h_s, 2, seed=11)
    t = time.time()
    blob, _ = ocodec.encode_video(spec, frames, 'LDP_1')
    ocodec.decode_video(spec, blob)
    dt = time.time() - t
    fps = 2.0 / dt / scale
    sample = ('oracle encode+decode of 2 frames (I+P, LDP_1) at %dx%d in %.1f s on %d threads'
              % (w_s, h_s, dt, cores))
    if scale != 1:
        sample += '; fps divided by %d (pixel ratio to %dx%d)' % (scale, width, height)
    return {'value': round(fps, 5), 'unit': 'frames/s', 'cores': cores, 'kind': 'port', 'sample': sample,
            'probe_conv_gflops': round(gflops, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--gop', type=str, default='1_GOP_32')
    ap.add_argument('--units', type=int, default=4, help='intra-period units per step per GPU (4 x 33 = the 128-frame clip of BASELINE configs[3])')
    ap.add_argument('--max-batch', type=int, default=8)
    ap.add_argument('--entropy-streams', type=int, default=4, help='decoder: concurrent range-coder chains')
    ap.add_argument('--entropy-lookahead', type=int, default=2, help='decoder: dependency levels of entropy decoding issued ahead')
    ap.add_argument('--tiny', action='store_true', help='tiny model widths (debug only; invalid as a result)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or bool(os.environ.get('AIVC_FORCE_DIST'))  # (the env var exercises the RCCL path on 1 GPU)
    if use_dist:
        dist.init_process_group('nccl', device_id=dev)

    from aivc_amd import ops, synth
    from aivc_amd.func_util.GOP_structure import generate_gop_struct
    from aivc_amd.models import arch
    from aivc_amd.parallel import broadcast_model
    widths = arch.TINY_WIDTHS if args.tiny else arch.DEFAULT_WIDTHS
    seed = 1234
    model = synth.make_model(widths, seed=seed, device=dev)
    synth.calibrate_operating_point(model, dev)
    if use_dist:
        broadcast_model(model)  # the one collective: weights over RCCL/xGMI
    from aivc_amd.codec import FrameCodec
    fc = FrameCodec(model, max_batch=args.max_batch, entropy_streams=args.entropy_streams,
                    entropy_lookahead=args.entropy_lookahead)
    unit = len(generate_gop_struct(args.gop))
    per_step = unit * args.units
    n_total = args.warmup + args.steps + 1
    # unit u of this rank's share = global unit (rank + u * world)
    clips = [gpu_synthetic_unit(args.width, args.height, per_step, (rank + i * world) * per_step, dev, 666 + rank + i * world)
             for i in range(n_total)]
    torch.cuda.synchronize()

    stats = {'enc_s': 0.0, 'dec_s': 0.0, 'bytes': 0}

    def step(i, timed=False):
        with torch.no_grad():
            t0 = time.time()
            enc = fc.encode_video(clips[i], args.gop)
            blob = fc.assemble_video(enc)
            if timed:
                torch.cuda.synchronize()
                t1 = time.time()
            dec, _, _, _ = fc.decode_video(blob, dev)
            if timed:
                torch.cuda.synchronize()
                stats['enc_s'] += t1 - t0
                stats['dec_s'] += time.time() - t1
                stats['bytes'] += len(blob)
        return enc, dec

    closed_loop = True
    for i in range(args.warmup):
        enc, dec = step(i)
        rec = [r for g in enc['recs'] for r in g][:len(dec)]
        closed_loop &= all(torch.equal(d[k], e[k]) for d, e in zip(dec, rec) for k in 'yuv')

    def barrier():
        if use_dist:
            dist.barrier()
    barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i, timed=True)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.time() - t0
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # quality of what was coded (outside the timed region; on-device CLIC metrics, aivc_amd/clic21): first
    # intra-period unit of the first timed clip.  With the synthetic random-init weights the figures say nothing
    # about the codec's rate-distortion -- they are the "PSNR/bpp" slots of the metric, filled by the same code
    # that would score real weights.
    quality = None
    if rank == 0 and not os.environ.get('AIVC_NO_QUALITY'):
        from aivc_amd.clic21.metrics import evaluate
        with torch.no_grad():
            q_enc = fc.encode_video(clips[args.warmup][:unit], args.gop)
            q_blob = fc.assemble_video(q_enc)
            q_dec, _, _, _ = fc.decode_video(q_blob, dev)
        target, submit = {}, {}
        for i, (src, d) in enumerate(zip(clips[args.warmup][:unit], q_dec)):
            for k in 'yuv':
                target['%d_%s' % (i, k)] = src[k]
                submit['%d_%s' % (i, k)] = d[k]
        r = evaluate(submit, target)
        quality = {'frames': unit, 'psnr_db': round(float(r['PSNR']), 4), 'ms_ssim': round(float(r['MSSSIM']), 6),
                   'bpp': round(len(q_blob) * 8.0 / (unit * args.width * args.height), 5),
                   'note': 'synthetic random-init weights: not a rate-distortion result'}

    roofline = None
    if not args.no_roofline and rank == 0:
        ops.PROFILE = []
        step(n_total - 1)
        torch.cuda.synchronize()
        per = {}
        shapes = {}
        for variant, flops, e0, e1, shape in ops.PROFILE:
            sec_ = e0.elapsed_time(e1) * 1e-3
            for table, key in ((per, variant), (shapes, (variant,) + shape)):
                d = table.setdefault(key, [0, 0.0, 0.0])
                d[0] += 1
                d[1] += flops
                d[2] += sec_
        if os.environ.get('AIVC_LAYER_TABLE'):  # tuning aid: per-layer-shape time of one step, on stderr
            for key, d in sorted(shapes.items(), key=lambda kv: -kv[1][2]):
                v, mode, k, st, ci, co, nb, hh, ww, g = key
                sys.stderr.write('%8.2f ms %5d x  %6.1f TF/s  %-30s mode%d k%d s%d %3d->%3d%s n%d %dx%d\n' % (
                    d[2] * 1e3, d[0], d[1] / d[2] / 1e12, VARIANT_NAMES.get(v, str(v)), mode, k, st, ci, co,
                    '+gdn' if g else '', nb, hh, ww))
        ops.PROFILE = None
        mf = {v: d for v, d in per.items() if v >= 100}
        if mf:
            dom = max(mf, key=lambda v: mf[v][2])
            cnt, fl, sec = mf[dom]
            all_fl = sum(d[1] for d in mf.values())
            all_sec = sum(d[2] for d in mf.values())
            traffic, traffic_note = None, None
            pmc = os.path.join(ROOT, 'profiles', 'r01_pmc_probe.json')
            if os.path.exists(pmc):
                pj = json.load(open(pmc))
                if pj.get('kernel') == VARIANT_NAMES.get(dom, str(dom)):
                    traffic = pj['traffic_bytes_per_launch']
                    traffic_note = ('HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on '
                                    'the probe shape (%s): %d B vs %d B algorithmic' % (pj['probe'], traffic,
                                                                                         pj['algorithmic_bytes_per_launch']))
            roofline = {'bound': 'mfma', 'kernel': VARIANT_NAMES.get(dom, str(dom)),
                        'achieved': round(fl / sec / 1e12, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(fl / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_note': traffic_note,
                        'launches': cnt, 'avg_launch_us': round(sec / cnt * 1e6, 2),
                        'gflop_per_launch': round(fl / cnt / 1e9, 3),
                        'all_mfma_conv': {'achieved': round(all_fl / all_sec / 1e12, 2),
                                          'frac': round(all_fl / all_sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                          'tflop_per_step': round(all_fl / 1e12, 3),
                                          'kernel_s_per_step': round(all_sec, 4)},
                        'per_variant': {VARIANT_NAMES.get(v, str(v)): {'launches': d[0], 'tflops': round(d[1] / d[2] / 1e12, 2),
                                                                       'ms_total': round(d[2] * 1e3, 2)}
                                        for v, d in sorted(per.items())}}

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.width, args.height, model)

    if rank == 0:
        frames = world * args.steps * per_step
        out = {
            'metric': 'encode+decode fps @1080p YUV420 (RA GOP32)',
            'value': round(frames / elapsed, 4), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 2), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%dx%d 8-bit YUV420, random access %s: %d-frame clip (%d intra-period units of %d frames) encoded + '
                                   'decoded per step per GPU; synthetic random-init stand-in for model ms_ssim-4 '
                                   '(widths %s); units sharded across GPUs' % (args.width, args.height, args.gop, per_step, args.units, unit, widths),
                       'frames_per_step': per_step, 'units_per_step': args.units, 'parallelism': 'unit-sharded x%d' % world},
            'encode_fps_rank0': round(args.steps * per_step / stats['enc_s'], 3),
            'decode_fps_rank0': round(args.steps * per_step / stats['dec_s'], 3),
            'bytes_per_frame': round(stats['bytes'] / (args.steps * per_step), 1),
            'closed_loop_ok': bool(closed_loop), 'quality': quality,
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        if args.tiny:
            out['invalid'] = 'tiny debug model'
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
