#!/usr/bin/env python3
"""bench.py -- encode+decode throughput of the AIVC hot path on MI355X.

One "step" = encode + decode of ONE 128-frame clip of synthetic 1920x1080 8-bit YUV 4:2:0 video under
random-access coding `1_GOP_32` (BASELINE.json configs[3], the configuration the metric is quoted on; it fits
one GPU): 4 intra-period units of 33 frames (I, P, 31 hierarchical B) = 132 coded frames, the last 4 being the
repetition of the last frame that completes the last GOP (src/model_mngt/model_management.py:142-153).
`value` counts the 128 REQUESTED frames over wall time, as the reference prints it (real_life/encode.py:165);
coded frames per second are reported next to it.  Inputs are resident in HBM before the timed region.

N GPUs (one process per GPU, torch.distributed over RCCL):
  --scaling strong (default for N > 1): the SAME clip on all GPUs -- units over min(N, 4) groups, temporal-layer
      sharding inside a group (aivc_amd/parallel.py: ClipShard); per level one all_gather of 8-bit frames inside
      the group, per clip the gather of the bitstream; bytes asserted identical to a single-rank encode.
  --scaling weak: every rank its own clip, no data-path collective at all; also measured in the default run
      and reported as `weak_scaling` (the replica figure).
One broadcast of the weights at start in both modes.

Prints ONE JSON line on rank 0 (see the driver contract) with extra objects:
  roofline      dominant kernel (fp32 MFMA implicit-GEMM conv): algorithmic FLOPs per launch / average launch
                duration, measured with HIP events on the launch stream during an extra, untimed, instrumented
                step; `hbm_stages`: achieved GB/s of the memory-bound stages against the 8 TB/s HBM peak.
  cpu_baseline  the CPU oracle (a port, oracle/) with its transforms on torch-CPU (what the reference's --cpu path
                spends its time in) timed on this box's host cores on a bounded sample (all cores and one core,
                encode and decode separately); `fmaf_oracle`: the parity checker on the same frames -- the GPU
                codes them with the same default-width model and the bytes / reconstructions are asserted equal
                (`parity_checked`).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import aivc_amd  # noqa: E402,F401  (first: sets the runtime's hardware-queue count before the GPU is touched)
import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0         # same guide: HBM3E 8 TB/s (spec; 6.3 TB/s is the measured copy ceiling)
FP64_VALU_PEAK_TFLOPS = 78.6  # fp64 vector rate = the unpacked fp32 vector rate (half of the guide's packed 157.3)
CDF_FLOP_PER_POINT = 70       # estimate, see `valu_stages`
_TILES = {0: '128x128', 1: '64x64', 2: '256x64', 3: '128x32', 4: '256x128', 5: '64x128', 6: '128x64'}
_MODES = {0: 'conv', 1: 'tconv', 2: 'gdn'}


def variant_name(v):
    """aivc_conv2d_variant code -> readable kernel name"""
    if v == 0:
        return 'conv_direct_kernel'
    if v == 1:
        return 'thin_tconv_kernel'
    if v == 2:
        return 'thin_mfma_kernel'
    if v == 190:
        return 'conv_mfma<conv,128x64+tail1x1>'
    if v == 301:
        return 'conv_wino<8x8 tiles,64>'
    if v == 302:
        return 'conv_wino<5x5s2 polyphase>'
    if v == 303:
        return 'conv_wino<tconv5x5s2 classes>'
    if v == 400:
        return 'gdn_resident'
    if v >= 1000:
        return 'bf16x3:' + variant_name(v - 1000)
    if v == 191:
        return 'conv_images<4x32px,64+gdn>'
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


def cpu_baseline(width, height, model, fc, dev, gop_name='1_GOP_32', unit_frames=33):
    """CPU legs on this box's host cores, encode and decode timed separately (the reference cannot run: its model
    sources, weights and torchac are absent):
      * `value`: the oracle with its transforms on torch-CPU (oracle/torch_cpu.py: F.conv2d / conv_transpose2d /
        GDN as the reference's --cpu path runs them, src/encode.py:85-93) on ONE FULL INTRA-PERIOD UNIT of the
        bench's coding structure at full frame size ("a reduced frame count, >= 1 full
        intra-period unit"; 33 frames of `1_GOP_32` at 1080p take ~90 s), thread count swept beforehand;
        `one_core`: an I + P + B triple (`1_GOP_2`) at full frame size on one thread;
      * `fmaf_oracle`: the parity checker itself (fixed-order fmaf chains, OpenMP) on the same full-size triple --
        the GPU codes the SAME frames with the same default-width model and bytes + reconstructions must be
        equal (`parity_checked`).  It is the checker, not a representative CPU implementation (~1 % of host peak)."""
    import numpy as np
    from aivc_amd import synth
    from oracle import codec as ocodec
    from oracle import oracle as orc
    from oracle import spec as ospec
    from oracle import torch_cpu
    orc.lib()
    from aivc_amd import abi as _abi, ops as _ops
    orc.set_precision('fp32w' if _ops.PRECISION == _abi.PREC_FP32_WINO else 'fp32')  # the checker walks the contract version the GPU runs
    cores = os.cpu_count() or 1
    spec = ospec.export_model(model)
    frames = synth.synthetic_video(width, height, 3, seed=11)

    def timed(fr, threads, gop='1_GOP_2'):
        with torch_cpu.torch_convs(threads):
            t0 = time.time()
            blob, recs = ocodec.encode_video(spec, fr, gop)
            t1 = time.time()
            dec = ocodec.decode_video(spec, blob)
            t2 = time.time()
        closed = all(np.array_equal(d[k], r[k]) for d, r in zip(dec, recs) for k in 'yuv')
        return t1 - t0, t2 - t1, closed, len(blob)

    # thread count: ATen / oneDNN on batch-1 convolutions does not scale to every hardware thread of a big host (all
    # 256 threads of the MI355X box measured 19x SLOWER than one thread per pixel: 306 s for the triple) -- the count is
    # picked by timing ONE INTRA FRAME AT THE FULL FRAME SIZE (round 6: the sweep ran on a 480x256 triple before and its
    # optimum, 16 threads, was then applied to 1080p), as a user of the reference's --cpu path would tune it
    w1, h1 = width, height
    small = synth.synthetic_video(w1, h1, 1, seed=11)
    timed(synth.synthetic_video(max(64, width // 4 // 16 * 16), max(48, height // 4 // 16 * 16), 1, seed=11), min(cores, 8), '1_GOP_0')  # thread pool / oneDNN primitive warm-up
    sweep = {}
    for t in sorted({cores, max(1, cores // 2), max(1, cores // 4), 64, 32, 16, 8}):
        if t <= cores:
            e_, d_, _, _ = timed(small, t, '1_GOP_0')
            sweep[t] = round(e_ + d_, 3)
            if e_ + d_ > 3 * min(sweep.values()):
                break  # (counts are tried in ascending order: past the optimum it only gets worse)
    best = min(sweep, key=sweep.get)
    # the timed sample: one whole unit of the bench's structure (AIVC_CPU_BASELINE_FRAMES trims it for quick runs)
    n_unit = int(os.environ.get('AIVC_CPU_BASELINE_FRAMES', unit_frames))
    unit_gop = gop_name if n_unit == unit_frames else '1_GOP_%d' % (n_unit - 1)
    unit = synth.synthetic_video(width, height, n_unit, seed=11)
    with torch_cpu.torch_convs(best):
        t0 = time.time()
        blob_u, recs_u = ocodec.encode_video(spec, unit, unit_gop)
        t1 = time.time()
        dec_u = ocodec.decode_video(spec, blob_u)
        t2 = time.time()
    enc_s, dec_s, nbytes = t1 - t0, t2 - t1, len(blob_u)
    closed = all(np.array_equal(d[k], r[k]) for d, r in zip(dec_u, recs_u) for k in 'yuv')
    del dec_u, recs_u
    out = {'value': round(n_unit / (enc_s + dec_s), 5), 'unit': 'frames/s', 'cores': best, 'host_threads': cores, 'kind': 'port',
           'encode_fps': round(n_unit / enc_s, 5), 'decode_fps': round(n_unit / dec_s, 5), 'closed_loop': bool(closed),
           'frames': n_unit, 'thread_sweep_s': {str(k): v for k, v in sweep.items()},
           'sample': 'oracle with torch-CPU transforms (F.conv2d / conv_transpose2d / GDN, torch.set_num_threads(%d): the fastest of '
                     'the counts swept on one intra frame at %dx%d, host has %d) on %d frames = one full intra-period unit of %s at %dx%d: '
                     'encode %.1f s, decode %.1f s (%d bytes)' % (best, w1, h1, cores, n_unit, unit_gop, width, height, enc_s, dec_s, nbytes)}
    e1, d1, c1, _ = timed(frames, 1)
    out['one_core'] = {'value': round(3.0 / (e1 + d1), 6), 'encode_fps': round(3.0 / e1, 6),
                       'decode_fps': round(3.0 / d1, 6), 'cores': 1, 'closed_loop': bool(c1), 'frames': 3,
                       'sample': '3 frames I+P+B (1_GOP_2) at %dx%d on 1 thread: encode %.1f s, decode %.1f s'
                                 % (width, height, e1, d1)}
    # ---- the parity checker on the same full-size frames, and the HIP product path against it
    x = np.random.default_rng(0).standard_normal((1, 135, 240, 128), dtype=np.float32)
    w = np.random.default_rng(1).standard_normal((128, 3, 3, 128), dtype=np.float32) * 0.03
    orc.conv2d(x[:, :16], w, None, pad=1)
    t = time.time()
    orc.conv2d(x, w, None, pad=1)
    gflops = 9.56 / max(time.time() - t, 1e-6)
    est_full = 6200.0 / gflops  # ~6.2 TFLOP for I + P + B encode + decode at 1080p with the default widths
    w_s, h_s = (width, height) if est_full <= 150.0 else (width // 2, height // 2)
    pfr = frames if (w_s, h_s) == (width, height) else synth.synthetic_video(w_s, h_s, 3, seed=11)
    t0 = time.time()
    blob, recs = ocodec.encode_video(spec, pfr, '1_GOP_2')
    t1 = time.time()
    dec = ocodec.decode_video(spec, blob)
    t2 = time.time()
    with torch.no_grad():
        g_enc = fc.encode_video(synth.to_device_frames(pfr, dev), '1_GOP_2')
        g_blob = fc.assemble_video(g_enc)
        g_dec, _, _, _ = fc.decode_video(g_blob, dev)
    parity = g_blob == blob and all(np.array_equal(g[k][0].cpu().numpy(), r[k]) and np.array_equal(r[k], d[k])
                                    for g, r, d in zip(g_dec, recs, dec) for k in 'yuv')
    if not parity:
        raise SystemExit('bench.py: HIP bitstream / reconstruction differs from the CPU oracle at default widths')
    scale = (width * height) / float(w_s * h_s)
    out.update(parity_checked=True, parity_bytes=len(blob),
               fmaf_oracle={'value': round(3.0 / (t2 - t0) / scale, 5), 'unit': 'frames/s', 'cores': cores,
                            'probe_conv_gflops': round(gflops, 1),
                            'sample': 'the parity checker (OpenMP + fmaf chains in the order of the arithmetic contract) on the '
                                      'triple at %dx%d: encode %.1f s, decode %.1f s' % (w_s, h_s, t1 - t0, t2 - t1)})
    return out


_REAL_STDOUT = 1


def _self_launch(n):
    """re-exec this script under torch.distributed.run with n ranks on this node"""
    import socket
    import subprocess
    if not os.environ.get('AIVC_BENCH_SINGLE_DEVICE') and torch.cuda.device_count() < n:
        raise SystemExit('bench.py: --gpus %d but only %d GPU(s) visible' % (n, torch.cuda.device_count()))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    os.dup2(_REAL_STDOUT, 1)  # the child job's rank 0 writes the one line there
    raise SystemExit(subprocess.call(cmd, env=env))


def _structure_label(gop_name):
    """'1_GOP_32' -> 'RA GOP32', 'LDP_8' -> 'LDP 8', '1_GOP_0' -> 'all intra' (src/func_util/GOP_structure.py:199-221)"""
    toks = gop_name.split('_')
    if toks[0] == 'LDP':
        return 'LDP %s' % toks[1]
    if toks[-1] == '0':
        return 'all intra'
    return 'RA GOP%s' % toks[-1] + ('' if toks[0] == '1' else ' x%s chained' % toks[0])


def main():
    # stdout carries exactly one line (the result); file descriptor 1 is pointed at stderr for everything else,
    # including native libraries that print there
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--gop', type=str, default='1_GOP_32')
    ap.add_argument('--frames', type=int, default=128, help='requested frames of the clip (BASELINE configs[3]: 128)')
    ap.add_argument('--scaling', choices=('auto', 'strong', 'weak'), default=os.environ.get('AIVC_BENCH_SCALING', 'auto'),
                    help='auto = strong (one clip over all GPUs) when N > 1')
    ap.add_argument('--max-batch', type=int, default=64, help='frames of one dependency level per launch (64 = the widest level of the clip in one batch: +0.9 %% over 16, same-box A/B)')
    ap.add_argument('--entropy-streams', type=int, default=8, help='decoder: concurrent range-coder chains')
    ap.add_argument('--entropy-lookahead', type=int, default=0, help='decoder: dependency levels of entropy decoding issued ahead (0: the whole clip up front)')
    ap.add_argument('--no-high-rate', action='store_true', help='skip the high-rate operating point (every y feature map coded) measured after the headline run')
    ap.add_argument('--high-rate-steps', type=int, default=3)
    ap.add_argument('--no-lean-encoder', action='store_true', help="skip the bitstream-only encoder (recon='refs') measured after the headline run (its own object, never `value`)")
    ap.add_argument('--lean-encoder-steps', type=int, default=2)
    ap.add_argument('--no-contract-v2', '--no-other-contract', dest='no_contract_v2', action='store_true', help="skip the other version of the fp32 contract (version 1 when the run is version 2 and vice versa) measured after the headline run (its own object `contract_v1` / `contract_v2`, never `value`)")
    ap.add_argument('--contract-v2-steps', type=int, default=2)
    ap.add_argument('--no-pipelined', action='store_true', help='skip the two-clips-in-flight schedule measured after the headline run (its own object, never `value`)')
    ap.add_argument('--pipelined-steps', type=int, default=3)
    ap.add_argument('--no-precision-mode', action='store_true', help='skip the bf16x3 precision mode measured after the headline run (its own object, never `value`)')
    ap.add_argument('--precision-steps', type=int, default=2)
    ap.add_argument('--contract', choices=('fp32', 'fp32w'), default=os.environ.get('AIVC_BENCH_CONTRACT', os.environ.get('AIVC_CONTRACT', 'fp32w')),
                    help="version of the fp32 arithmetic contract the run computes in (default: the library's, aivc_amd/ops.py DEFAULT_CONTRACT): 'fp32' = version 1 (tap chains), 'fp32w' = version 2 (Winograd F(2x2,3x3) chains for the stride-1 3x3 layers it covers, include/aivc_hip.h); HIP == CPU oracle bit for bit in both; the other version is measured beside the headline (`contract_v1` / `contract_v2`)")
    ap.add_argument('--widths', type=str, default='default',
                    help="model widths: 'default' (n2 64 / n 128: the stand-in the headline is quoted on), 'w192' (n2 96 / n 192 / c_y 96), 'w144' "
                         "(n2 72 / n 144 / c_y 72: not multiples of 32), or 'n2=..,n=..,c_y=..,c_short=..,c_z=..,n_h=..'; anything but 'default' is a "
                         "side measurement (the real widths are unknown, SURVEY F1/F2)")
    ap.add_argument('--tiny', action='store_true', help='tiny model widths (debug only; invalid as a result)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--active-y', type=str, default='6,12',
                    help='non-zero y feature maps MOFNet,CodecNet the synthetic model is calibrated to (64,64 = high rate)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL), exactly the
        # command the driver uses; rank 0 of the child job prints the line on the stdout we hand down
        return _self_launch(args.gpus)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but launched with WORLD_SIZE=%d' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    # validation aid for boxes with ONE GPU: AIVC_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # AIVC_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks per device) -- the whole N > 1 flow of this
    # script (sharding, collectives, both scaling modes) then runs for real across processes; not a measurement
    if os.environ.get('AIVC_BENCH_SINGLE_DEVICE'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or bool(os.environ.get('AIVC_FORCE_DIST'))  # (the env var exercises the RCCL path on 1 GPU)
    backend = os.environ.get('AIVC_DIST_BACKEND', 'nccl')
    if use_dist:
        if 'RANK' not in os.environ:  # AIVC_FORCE_DIST=1 python bench.py: a one-rank job of its own
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                                  MASTER_PORT=str(sk.getsockname()[1]))
        from aivc_amd import parallel as _par
        _par.init_process_group(dev, backend)  # RCCL bound to this rank's GPU, finite timeout, watchdog
    strong = args.scaling == 'strong' or (args.scaling == 'auto')  # the label: same total work whatever N
    sharded = strong and use_dist  # ClipShard path; at N = 1 strong and weak are the same single-process run

    from aivc_amd import ops, parallel, synth
    from aivc_amd.func_util.GOP_structure import generate_gop_struct
    ops.set_precision(args.contract)
    from aivc_amd.models import arch
    presets = {'default': arch.DEFAULT_WIDTHS,
               'w192': {'n2': 96, 'n': 192, 'c_y': 96, 'c_short': 96, 'c_z': 48, 'n_h': 192},
               'w144': {'n2': 72, 'n': 144, 'c_y': 72, 'c_short': 72, 'c_z': 36, 'n_h': 144}}
    if args.widths in presets:
        widths = presets[args.widths]
    else:
        widths = dict(arch.DEFAULT_WIDTHS, **{k: int(v) for k, v in (kv.split('=') for kv in args.widths.split(','))})
    if args.tiny:
        widths = arch.TINY_WIDTHS
    seed = 1234
    model = synth.make_model(widths, seed=seed, device=dev)
    active_y = tuple(int(v) for v in args.active_y.split(','))
    synth.calibrate_operating_point(model, dev, active_y=active_y)
    if use_dist:
        parallel.broadcast_model(model)  # the one collective on the weights: RCCL over xGMI
    from aivc_amd.codec import FrameCodec
    fc = FrameCodec(model, max_batch=args.max_batch, entropy_streams=args.entropy_streams,
                    entropy_lookahead=args.entropy_lookahead)
    unit = len(generate_gop_struct(args.gop))
    n_units = -(-args.frames // unit)
    coded = n_units * unit

    def make_clip(index):
        """clip `index`: args.frames distinct frames, the last unit completed by repeating the last frame"""
        fr = gpu_synthetic_unit(args.width, args.height, args.frames, index * args.frames, dev, 666 + index)
        fr = fr + [fr[-1]] * (coded - args.frames)
        return [fr[u * unit:(u + 1) * unit] for u in range(n_units)]

    n_total = args.warmup + args.steps + 1
    shard = parallel.ClipShard(n_units, dev) if sharded else None
    # strong: every rank holds the same clips; weak: rank r codes clips r, r + world, ...
    clips = [make_clip(i if strong else rank + i * world) for i in range(n_total)]
    torch.cuda.synchronize()
    stats = {'enc_s': 0.0, 'dec_s': 0.0, 'bytes': 0, 'events': []}

    def step(clip, timed=False, sh=None):
        """encode then decode one clip -> (gop records, data_dim, {unit: reconstructions of the encoder}, decoded).
        No host synchronisation inside a timed step: the decoder's entropy stage (side streams; it needs the bitstream
        only) starts while the main stream still runs the encoder's last synthesis.  The encode / decode split of a
        step is read from events on the main stream after the run."""
        with torch.no_grad():
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if sh is not None:
                blobs, dd = parallel.encode_clip(fc, clip, args.gop, shard=sh)
                enc_recs = None
            else:
                blobs, enc_recs, dd = fc.encode_units(clip, args.gop)
            if timed:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
            if sh is not None:
                dec = parallel.decode_clip(fc, blobs, dd, dev, shard=sh)
            else:
                dec = dict(enumerate(fc.decode_units(blobs, dd, dev)))
            if timed:
                e2 = torch.cuda.Event(enable_timing=True)
                e2.record()
                stats['events'].append((e0, e1, e2))
                stats['bytes'] += sum(len(b) for b in blobs)
        return blobs, dd, enc_recs, dec

    def barrier():
        if use_dist:
            dist.barrier()

    def timed_run(sh, first):
        barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(first, first + args.steps):
            step(clips[i], timed=True, sh=sh)
        torch.cuda.synchronize()
        barrier()
        el = time.time() - t0
        for e0, e1, e2 in stats['events']:  # main-stream time of the two halves of every step
            stats['enc_s'] += e0.elapsed_time(e1) * 1e-3
            stats['dec_s'] += e1.elapsed_time(e2) * 1e-3
        del stats['events'][:]
        if use_dist:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    # ---- warm-up; closed loop (decoder == encoder reconstruction) and, for the sharded path, bytes == single rank
    def warm_up():
        closed, equal = True, None
        for i in range(args.warmup):
            blobs, dd, enc_recs, dec = step(clips[i], sh=shard)
            if enc_recs is None:  # sharded: this rank's own single-process encode of the same clip is the reference
                with torch.no_grad():
                    ref_blobs, enc_recs, _ = fc.encode_units(clips[i], args.gop)
                equal = (equal is not False) and blobs == ref_blobs
            for u, frs in dec.items():
                closed &= all(torch.equal(d[k], e[k]) for d, e in zip(frs, enc_recs[u]) for k in 'yuv')
        if use_dist:  # the verdict of every rank
            t = torch.tensor([0 if (closed and equal is not False) else 1], device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(t)
            if int(t.item()):
                equal = False if equal is not None else equal
                closed = closed and equal is not None
        return closed, equal

    # Levels with fewer frames than a group has ranks can be coded in row bands over the group (aivc_amd/bands.py); over
    # RCCL that transport (point-to-point halo exchange on a split sub-group) is taken only after THIS run has verified
    # it: the warm-up clip must come out byte-identical to the single-process encode, otherwise bands are switched off
    # on every rank and the warm-up is repeated.  (A first-contact hang is named by tools/rccl_preflight.py --codec.)
    row_bands = None
    if shard is not None and shard.R > 1 and os.environ.get('AIVC_BAND_LEVELS') is None and args.warmup > 0 \
            and (shard.R >= 4 or args.width * args.height >= 6000000):
        if args.contract == 'fp32':
            shard.band_levels, row_bands = True, 'on (warm-up clip byte-identical to the single-process encode)'
        else:  # (FrameCodec._banded refuses: a Winograd chain depends on the tile grid and size of the tensor it is computed in)
            row_bands = 'off: row bands exist under version 1 of the arithmetic contract only (--contract fp32); levels narrower than the group are coded by its first ranks'
    closed_loop, bytes_equal = warm_up()
    if row_bands and row_bands.startswith('on') and not (closed_loop and bytes_equal is not False):
        shard.band_levels, row_bands = False, 'switched off: the warm-up clip coded in row bands differed from the single-process encode'
        closed_loop, bytes_equal = warm_up()
    if bytes_equal is False:
        raise SystemExit('bench.py: sharded bitstream differs from the single-rank bitstream')
    elapsed = timed_run(shard, args.warmup)
    main_stats = dict(stats)
    clips_done_for_hr = args.steps * args.frames
    # every range-decode launch of the run left its bit count behind (real_life/bitstream.py): judged now, outside the
    # timed region -- a section that did not decode to where its payload ends would make closed_loop_ok meaningless
    stream_errors = len(fc.stream_errors())

    # ---- the other scaling mode, reported next to the headline (N > 1 only: at N = 1 they coincide)
    other = None
    if world > 1 and args.scaling == 'auto':
        weak_clips = [make_clip(rank + i * world) for i in range(args.steps)]
        clips_backup, clips = clips, weak_clips
        stats.update(enc_s=0.0, dec_s=0.0, bytes=0)
        el_w = timed_run(None, 0)
        clips = clips_backup
        other = {'scaling': 'weak', 'value': round(world * args.steps * args.frames / el_w, 4), 'unit': 'frames/s',
                 'ms_per_step': round(el_w / args.steps * 1e3, 2),
                 'note': 'every rank its own %d-frame clip (replicas, no data-path collective)' % args.frames}
        del weak_clips
    stats = main_stats

    # quality of what was coded (outside the timed region; on-device CLIC metrics, aivc_amd/clic21): first
    # intra-period unit of the first timed clip.  With the synthetic random-init weights the figures say nothing
    # about the codec's rate-distortion -- they are the "PSNR/bpp" slots of the metric, filled by the same code
    # that would score real weights.
    quality = None
    if rank == 0 and not os.environ.get('AIVC_NO_QUALITY'):
        from aivc_amd.clic21.metrics import evaluate
        src_unit = clips[args.warmup][0]
        with torch.no_grad():
            q_blobs, _, q_dd = fc.encode_units([src_unit], args.gop)
            q_dec = fc.decode_units(q_blobs, q_dd, dev)[0]
        target, submit = {}, {}
        for i, (src, d) in enumerate(zip(src_unit, q_dec)):
            for k in 'yuv':
                target['%d_%s' % (i, k)] = src[k]
                submit['%d_%s' % (i, k)] = d[k]
        r = evaluate(submit, target)
        quality = {'frames': unit, 'psnr_db': round(float(r['PSNR']), 4), 'ms_ssim': round(float(r['MSSSIM']), 6),
                   'bpp': round(len(q_blobs[0]) * 8.0 / (unit * args.width * args.height), 5),
                   'note': 'synthetic random-init weights: not a rate-distortion result'}

    roofline = None
    if not args.no_roofline and rank == 0:
        ops.PROFILE = []
        ops.PROFILE_HBM = []
        ops.PROFILE_DIRECT_EQUIVALENT[:] = [0.0, 0.0]
        step(clips[n_total - 1])  # one single-rank step, instrumented
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
        hbm = {}
        for name, nbytes, e0, e1 in ops.PROFILE_HBM:
            d = hbm.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += nbytes
            d[2] += e0.elapsed_time(e1) * 1e-3
        if os.environ.get('AIVC_LAYER_TABLE'):  # tuning aid: per-layer-shape time of one step, on stderr
            for key, d in sorted(shapes.items(), key=lambda kv: -kv[1][2]):
                v, mode, k, st, ci, co, nb, hh, ww, g = key
                sys.stderr.write('%8.2f ms %5d x  %6.1f TF/s  %-30s mode%d k%d s%d %3d->%3d%s n%d %dx%d\n' % (
                    d[2] * 1e3, d[0], d[1] / d[2] / 1e12, VARIANT_NAMES.get(v, str(v)), mode, k, st, ci, co,
                    '+gdn' if g else '', nb, hh, ww))
        ops.PROFILE = None
        ops.PROFILE_HBM = None
        mf = {v: d for v, d in per.items() if v >= 100}
        if mf:
            dom = max(mf, key=lambda v: mf[v][2])
            cnt, fl, sec = mf[dom]
            all_fl = sum(d[1] for d in mf.values())
            all_sec = sum(d[2] for d in mf.values())
            traffic, traffic_note, counters = None, None, None
            # counter passes on this shape (tools/pmc_conv.sh; separate --pmc runs, never inside this process): the newest round's
            # (the newest summary under profiles/ whose `kernel` is this run's dominant kernel: r0*_pmc_dominant.json of the tap
            # kernels, tools/pmc_summary.py's r0*_pmc_<tag>.json of the Winograd / GDN kernels)
            cands = []
            for pmc in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_pmc_*.json'))):
                try:
                    pj = json.load(open(pmc))
                except ValueError:
                    continue
                if isinstance(pj, dict) and pj.get('kernel') == VARIANT_NAMES.get(dom, str(dom)) and pj.get('traffic_bytes_per_launch'):
                    cands.append((os.path.basename(pmc)[:3], pj.get('algorithmic_bytes_per_launch') or 0, pmc, pj))
            if cands:
                _, _, pmc, pj = max(cands, key=lambda c: (c[0], c[1]))  # the newest round's, its largest probe
                traffic = pj.get('traffic_bytes_per_launch')
                traffic_note = ('separate rocprofv3 --pmc passes on ONE launch shape of this kernel (%s): %d B per launch (FETCH_SIZE doubled per the guide) '
                                'against %s B algorithmic (%sx), L2 hit rate %s; %s' % (
                                    pj.get('probe', '?'), traffic, pj.get('algorithmic_bytes_per_launch'), pj.get('traffic_over_algorithmic'),
                                    pj.get('l2_hit_rate'), pj.get('note', os.path.basename(pmc))))
                counters = pj.get('sq_counters')
            roofline = {'bound': 'mfma', 'kernel': VARIANT_NAMES.get(dom, str(dom)),
                        'achieved': round(fl / sec / 1e12, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(fl / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_note': traffic_note,
                        'mfma_counters': counters,
                        'launches': cnt, 'avg_launch_us': round(sec / cnt * 1e6, 2),
                        'gflop_per_launch': round(fl / cnt / 1e9, 3),
                        # a Winograd instantiation is priced on the FLOPs it EXECUTES (what the matrix pipe is busy with); beside it the
                        # rate on the tap-chain work of the same layers (36 / 16 of it for the 3x3 form, 100 / 49 for the 5x5 forms):
                        # what the direct algorithm would have had to sustain for the same launch times -- above the peak by construction
                        'flops_priced': 'executed' if dom in (301, 302, 303) else 'algorithmic',
                        'tap_chain_equivalent': ({'achieved': round(fl / sec / 1e12 * (2.25 if dom == 301 else 100.0 / 49.0), 2),
                                                  'frac': round(fl / sec / 1e12 * (2.25 if dom == 301 else 100.0 / 49.0) / MFMA_F32_PEAK_TFLOPS, 4)}
                                                 if dom in (301, 302, 303) else None),
                        'all_mfma_conv': {'achieved': round(all_fl / all_sec / 1e12, 2),
                                          'frac': round(all_fl / all_sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                          'tflop_per_step': round(all_fl / 1e12, 3),
                                          'kernel_s_per_step': round(all_sec, 4),
                                          # version 2 of the contract: the Winograd launches are priced on the FLOPs they execute;
                                          # this is the tap-chain work of the same layers (version 1 would have issued it)
                                          'winograd_replaces_tflop_per_step': round(ops.PROFILE_DIRECT_EQUIVALENT[0] / 1e12, 3),
                                          'winograd_executes_tflop_per_step': round(ops.PROFILE_DIRECT_EQUIVALENT[1] / 1e12, 3)},
                        'per_variant': {VARIANT_NAMES.get(v, str(v)): {'launches': d[0], 'tflops': round(d[1] / d[2] / 1e12, 2),
                                                                       'frac': round(d[1] / d[2] / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                                                       'ms_total': round(d[2] * 1e3, 2)}
                                        for v, d in sorted(per.items())},
                        'hbm_stages': {name: {'bound': 'hbm', 'launches': d[0], 'achieved': round(d[1] / d[2] / 1e9, 1),
                                              'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(d[1] / d[2] / 1e9 / HBM_PEAK_GBS, 4),
                                              'algorithmic_mb_per_launch': round(d[1] / d[0] / 1e6, 2),
                                              'ms_total': round(d[2] * 1e3, 2)}
                                       for name, d in sorted(hbm.items()) if not name.startswith('cdf_points:')},
                        # the Laplace CDF build: one fp64 expm1 polynomial per CDF point (include/aivc_detmath.h,
                        # ~35 fp64 fma = 70 FLOP with both branches of the divergent range split) -- bound by
                        # the fp64 vector rate (78.6 TFLOP/s: 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz), not by HBM
                        'valu_stages': {name.split(':', 1)[1]: {'bound': 'fp64_valu', 'launches': d[0],
                                                                'gpoints_per_s': round(d[1] / d[2] / 1e9, 2),
                                                                'est_flop_per_point': CDF_FLOP_PER_POINT,
                                                                'achieved': round(d[1] * CDF_FLOP_PER_POINT / d[2] / 1e12, 2),
                                                                'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                                                'frac': round(d[1] * CDF_FLOP_PER_POINT / d[2] / 1e12 / FP64_VALU_PEAK_TFLOPS, 4),
                                                                'ms_total': round(d[2] * 1e3, 2)}
                                        for name, d in sorted(hbm.items()) if name.startswith('cdf_points:')}}

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.width, args.height, model, fc, dev, args.gop, unit)

    # ---- two clips in flight (never the headline): the decoder's entropy stage depends on the bitstream only, so the decode of
    # clip i is split (FrameCodec.decode_units_begin / _finish) around the ENCODE of clip i + 1: the serial range-coder streams of
    # clip i (the I frames' y streams at the head) decode on the side streams under that encode instead of in front of the first
    # synthesis -- the schedule of a deployment, where encoder and decoder are two processes working on consecutive clips.  Same
    # kernels, same bytes; one clip more of latency.  `value` keeps the strictly serial step (encode, then decode, of one clip).
    def pipelined_run(codec, n_steps):
        with torch.no_grad():
            # correctness of the schedule first: clip 0's decode finished behind clip 1's encode == its encoder's reconstruction
            b0, r0, dd0 = codec.encode_units(clips[0], args.gop)
            h0 = codec.decode_units_begin(b0, dd0, dev)
            b1, r1, dd1 = codec.encode_units(clips[1 % len(clips)], args.gop)
            d0 = codec.decode_units_finish(h0)
            d1 = codec.decode_units(b1, dd1, dev)
            ok = all(torch.equal(d[k], e[k]) for dec_, enc_ in ((d0, r0), (d1, r1)) for du, eu in zip(dec_, enc_) for d, e in zip(du, eu) for k in 'yuv')
            ok = ok and len(codec.stream_errors()) == 0
            del d0, d1, r0, r1
            torch.cuda.synchronize()
            t0 = time.time()
            handle = None
            for i in range(n_steps):
                blobs, _, dd = codec.encode_units(clips[(args.warmup + i) % len(clips)], args.gop)
                if handle is not None:
                    codec.decode_units_finish(handle)
                handle = codec.decode_units_begin(blobs, dd, dev)
            codec.decode_units_finish(handle)
            torch.cuda.synchronize()
            el = time.time() - t0
        return {'value': round(n_steps * args.frames / el, 4), 'unit': 'frames/s', 'steps': n_steps, 'clips_in_flight': 2,
                'ms_per_step': round(el / n_steps * 1e3, 2), 'closed_loop_ok': bool(ok),
                'vs_headline': round(n_steps * args.frames / el / (clips_done_for_hr / elapsed), 4)}

    pipelined = None
    if rank == 0 and world == 1 and not args.no_pipelined:
        pipelined = pipelined_run(fc, args.pipelined_steps)
        pipelined['note'] = ('decode of clip i split around the encode of clip i + 1 (decode_units_begin / _finish): the entropy stage of a clip runs '
                             'under the next clip\'s transforms; the serial step stays the headline')

    # ---- the high-rate operating point (BASELINE configs[4] names a high-rate model: every y feature map of both
    # networks non-zero, the serial range coder's streams at their longest), same clip, same run, outside `value`
    high_rate = None
    c_y = widths['c_y']
    if rank == 0 and world == 1 and not args.no_high_rate and tuple(active_y) != (c_y, c_y):
        model_hr = synth.make_model(widths, seed=seed, device=dev)
        synth.calibrate_operating_point(model_hr, dev, active_y=(c_y, c_y))
        fc_hr = FrameCodec(model_hr, max_batch=args.max_batch, entropy_streams=args.entropy_streams,
                           entropy_lookahead=args.entropy_lookahead)
        with torch.no_grad():
            blobs, enc_recs, dd = fc_hr.encode_units(clips[0], args.gop)
            dec = fc_hr.decode_units(blobs, dd, dev)
            hr_closed = all(torch.equal(d[k], e[k]) for du, eu in zip(dec, enc_recs) for d, e in zip(du, eu) for k in 'yuv')
            hr_errs = len(fc_hr.stream_errors())
            del dec, enc_recs
            torch.cuda.synchronize()
            t0 = time.time()
            hr_bytes, evs = 0, []
            for i in range(args.high_rate_steps):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                blobs, _, dd = fc_hr.encode_units(clips[(args.warmup + i) % len(clips)], args.gop)
                ev[1].record()
                fc_hr.decode_units(blobs, dd, dev)
                ev[2].record()
                evs.append(ev)
                hr_bytes += sum(len(b) for b in blobs)
            torch.cuda.synchronize()
            el_hr = time.time() - t0
        enc_hr = sum(e[0].elapsed_time(e[1]) for e in evs) * 1e-3
        dec_hr = sum(e[1].elapsed_time(e[2]) for e in evs) * 1e-3
        high_rate = {'value': round(args.high_rate_steps * args.frames / el_hr, 4), 'unit': 'frames/s', 'steps': args.high_rate_steps,
                     'ms_per_step': round(el_hr / args.high_rate_steps * 1e3, 2),
                     'nonzero_y_maps': {'mofnet': c_y, 'codecnet': c_y, 'of': c_y},
                     'bytes_per_frame': round(hr_bytes / (args.high_rate_steps * coded), 1),
                     'encode_main_stream_fps': round(args.high_rate_steps * args.frames / enc_hr, 3),
                     'decode_main_stream_fps': round(args.high_rate_steps * args.frames / dec_hr, 3),
                     'closed_loop_ok': bool(hr_closed), 'stream_errors': hr_errs,
                     'vs_headline': round(args.high_rate_steps * args.frames / el_hr / (clips_done_for_hr / elapsed), 4),
                     'note': 'same clip and code path with the synthetic model calibrated so that every y feature map of both '
                             'networks is coded: each frame carries two serial range-coder streams of h_y*w_y*%d symbols' % c_y}
        if not args.no_pipelined:
            high_rate['pipelined'] = pipelined_run(fc_hr, args.pipelined_steps)
        del model_hr, fc_hr

    # ---- bitstream-only encoder (FrameCodec.encode_units(recon='refs'); never the headline: `value` keeps the encoder that
    # reconstructs every frame, as the reference's does): frames no other frame references skip their CodecNet synthesis
    lean_encoder = None
    if rank == 0 and world == 1 and not args.no_lean_encoder:
        with torch.no_grad():
            ref_blobs, _, _ = fc.encode_units(clips[0], args.gop)
            blobs, lean_recs, dd = fc.encode_units(clips[0], args.gop, recon='refs')
            le_same = blobs == ref_blobs
            le_skipped = sum(r is None for u in lean_recs for r in u)
            dec = fc.decode_units(blobs, dd, dev)
            le_closed = all(torch.equal(d[k], e[k]) for du, eu in zip(dec, lean_recs) for d, e in zip(du, eu) if e is not None for k in 'yuv')
            le_errs = len(fc.stream_errors())
            del dec, lean_recs, ref_blobs
            torch.cuda.synchronize()
            t0 = time.time()
            evs = []
            for i in range(args.lean_encoder_steps):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                blobs, _, dd = fc.encode_units(clips[(args.warmup + i) % len(clips)], args.gop, recon='refs')
                ev[1].record()
                fc.decode_units(blobs, dd, dev)
                ev[2].record()
                evs.append(ev)
            torch.cuda.synchronize()
            el_le = time.time() - t0
        enc_le = sum(e[0].elapsed_time(e[1]) for e in evs) * 1e-3
        dec_le = sum(e[1].elapsed_time(e[2]) for e in evs) * 1e-3
        lean_encoder = {'value': round(args.lean_encoder_steps * args.frames / el_le, 4), 'unit': 'frames/s', 'steps': args.lean_encoder_steps,
                        'ms_per_step': round(el_le / args.lean_encoder_steps * 1e3, 2),
                        'encode_main_stream_fps': round(args.lean_encoder_steps * args.frames / enc_le, 3),
                        'decode_main_stream_fps': round(args.lean_encoder_steps * args.frames / dec_le, 3),
                        'frames_not_reconstructed_by_the_encoder': '%d of %d coded' % (le_skipped, coded),
                        'bytes_equal_full_encoder': bool(le_same), 'closed_loop_ok_on_references': bool(le_closed), 'stream_errors': le_errs,
                        'vs_headline': round(args.lean_encoder_steps * args.frames / el_le / (clips_done_for_hr / elapsed), 4),
                        'note': "encode_units(recon='refs'): an encoder whose product is the bitstream -- the frames of the last dependency level "
                                '(no other frame references them) skip the CodecNet synthesis the headline encoder runs for its PSNR print; '
                                'same container bytes, the decoder is unchanged; reported beside the headline, never as `value`'}

    # ---- the OTHER version of the fp32 contract, beside the headline (never `value`).  Version 2 (AIVC_PREC_FP32_WINO,
    # 'fp32w', the library's default since round 6): the stride-1 3x3 layers with c_out % 128 == 0 and >= AIVC_WINO_MIN_PIXELS input
    # pixels on Winograd F(2x2, 3x3) chains -- still fixed-order fp32 chains, HIP == CPU oracle bit for bit
    # (tests/test_gpu_winograd.py), other bits than version 1 (the 9-tap chains, 'fp32').
    contract_v2 = None
    other_contract = 'fp32' if args.contract == 'fp32w' else 'fp32w'
    if rank == 0 and world == 1 and not args.no_contract_v2:
        prev_prec = ops.set_precision(other_contract)
        try:
            with torch.no_grad():
                blobs, enc_recs, dd = fc.encode_units(clips[0], args.gop)
                dec = fc.decode_units(blobs, dd, dev)
                v2_closed = all(torch.equal(d[k], e[k]) for du, eu in zip(dec, enc_recs) for d, e in zip(du, eu) for k in 'yuv')
                v2_errs = len(fc.stream_errors())
                del dec, enc_recs
                ops.PROFILE = []
                ops.PROFILE_DIRECT_EQUIVALENT[:] = [0.0, 0.0]
                torch.cuda.synchronize()
                t0 = time.time()
                for i in range(args.contract_v2_steps):
                    blobs, _, dd = fc.encode_units(clips[(args.warmup + i) % len(clips)], args.gop)
                    fc.decode_units(blobs, dd, dev)
                torch.cuda.synchronize()
                el_v2 = time.time() - t0
            wino = [0, 0.0, 0.0]
            for variant, flops, e0, e1, _shape in ops.PROFILE:
                if variant in (301, 302, 303):
                    wino[0] += 1
                    wino[1] += flops
                    wino[2] += e0.elapsed_time(e1) * 1e-3
            replaced, executed = ops.PROFILE_DIRECT_EQUIVALENT
            ops.PROFILE = None
        finally:
            ops.PROFILE = None
            ops.set_precision(prev_prec)
        contract_v2 = {
            'contract': other_contract, 'dtype': 'f32', 'value': round(args.contract_v2_steps * args.frames / el_v2, 4), 'unit': 'frames/s',
            'steps': args.contract_v2_steps, 'ms_per_step': round(el_v2 / args.contract_v2_steps * 1e3, 2),
            'vs_headline': round(args.contract_v2_steps * args.frames / el_v2 / (clips_done_for_hr / elapsed), 4),
            'closed_loop_ok': bool(v2_closed), 'stream_errors': v2_errs,
            'note': 'the other version of the fp32 arithmetic contract (include/aivc_hip.h; version 1 = 9-tap chains everywhere, version 2 = '
                    'AIVC_PREC_FP32_WINO: the covered 3x3 layers on Winograd F(2x2,3x3) chains, a fused GDN behind a covered layer as a second '
                    'launch): same code path and clip; each version is bit exact against the CPU oracle of the same version, their bits differ '
                    '(encoder and decoder must agree); reported beside the headline, never as `value`'}
        if other_contract == 'fp32w':
            contract_v2['winograd_kernel'] = {
                'launches_per_step': wino[0] // max(args.contract_v2_steps, 1),
                'ms_per_step': round(wino[2] / args.contract_v2_steps * 1e3, 2),
                # executed FLOPs (16 multiplications per 2 x 2 outputs and channel pair) over HIP-event time: the
                # matrix pipe's rate, priced against the fp32 MFMA peak
                'bound': 'mfma', 'achieved': round(wino[1] / max(wino[2], 1e-9) / 1e12, 2), 'peak': MFMA_F32_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(wino[1] / max(wino[2], 1e-9) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                'tap_chain_tflop_replaced_per_step': round(replaced / args.contract_v2_steps / 1e12, 3),
                'tflop_executed_per_step': round(executed / args.contract_v2_steps / 1e12, 3)}

    # ---- the bf16x3 precision MODE (aivc_conv_params.precision; never the headline: `value` stays the fp32 contract):
    # same clip, same model, same code path with the wide convolutions on six bf16 MFMA products per fp32 product
    precision_mode = None
    if rank == 0 and world == 1 and not args.no_precision_mode:
        prev_prec = ops.set_precision('bf16x3')
        try:
            with torch.no_grad():
                blobs, enc_recs, dd = fc.encode_units(clips[0], args.gop)
                dec = fc.decode_units(blobs, dd, dev)
                pm_closed = all(torch.equal(d[k], e[k]) for du, eu in zip(dec, enc_recs) for d, e in zip(du, eu) for k in 'yuv')
                pm_errs = len(fc.stream_errors())
                del dec, enc_recs
                ops.PROFILE = []
                torch.cuda.synchronize()
                t0 = time.time()
                for i in range(args.precision_steps):
                    blobs, _, dd = fc.encode_units(clips[(args.warmup + i) % len(clips)], args.gop)
                    fc.decode_units(blobs, dd, dev)
                torch.cuda.synchronize()
                el_pm = time.time() - t0
            per = {}
            for variant, flops, e0, e1, _shape in ops.PROFILE:
                d = per.setdefault(variant, [0, 0.0, 0.0])
                d[0] += 1
                d[1] += flops
                d[2] += e0.elapsed_time(e1) * 1e-3
            ops.PROFILE = None
            bf = {v: d for v, d in per.items() if v >= 1000}
            fp = {v: d for v, d in per.items() if 100 <= v < 1000}
        finally:
            ops.set_precision(prev_prec)
        precision_mode = {
            'dtype': 'bf16x3', 'value': round(args.precision_steps * args.frames / el_pm, 4), 'unit': 'frames/s',
            'steps': args.precision_steps, 'ms_per_step': round(el_pm / args.precision_steps * 1e3, 2),
            'vs_headline': round(args.precision_steps * args.frames / el_pm / (clips_done_for_hr / elapsed), 4),
            'closed_loop_ok': bool(pm_closed), 'stream_errors': pm_errs,
            # algorithmic fp32-equivalent FLOPs (2 per multiply-add of the layer, as for the fp32 kernels) over HIP-event
            # time; each fp32 product is six bf16 MFMA products, priced against the dense bf16 peak / 6
            'bf16x3_kernels': {'launches': sum(d[0] for d in bf.values()), 'ms_per_step': round(sum(d[2] for d in bf.values()) / args.precision_steps * 1e3, 2),
                               'fp32_equivalent_tflops': round(sum(d[1] for d in bf.values()) / max(sum(d[2] for d in bf.values()), 1e-9) / 1e12, 2),
                               'peak_fp32_equivalent_tflops': round(2500.0 / 6.0, 1),
                               'per_variant': {VARIANT_NAMES.get(v, str(v)): {'launches': d[0], 'fp32_equivalent_tflops': round(d[1] / d[2] / 1e12, 2),
                                                                              'ms_total': round(d[2] * 1e3, 2)} for v, d in sorted(bf.items())}},
            'fp32_contract_kernels_left': {'launches': sum(d[0] for d in fp.values()), 'ms_per_step': round(sum(d[2] for d in fp.values()) / args.precision_steps * 1e3, 2)},
            'note': 'precision MODE, not the headline: fp32 operands split exactly into three bf16 terms, six bf16 MFMA products per '
                    'fp32 product, fp32 accumulation, for the conv / transposed conv layers with c_in % 32 == 0 and c_out of 64 / 128 '
                    'and a reduction of 512 terms or more, weights split once per layer (the image layers, the thin output layer, the 1x1 convs '
                    'and the second GEMMs of the fused GDN / 1x1 tail stay on the fp32 contract); '
                    'within fp32 summation-order noise of the contract, not its bits (tests/test_gpu_precision.py); encoder and '
                    'decoder must run the same mode'}

    if rank == 0:
        # N = 1: strong and weak are the same single-process run, the label says so
        scaling = 'single' if world == 1 else ('strong' if strong else 'weak')
        clips_done = args.steps * (1 if strong else world)
        g = shard.G if shard is not None else 1
        r_ = shard.R if shard is not None else 1
        out = {
            'metric': 'encode+decode fps @%dp YUV420 (%s)' % (args.height, _structure_label(args.gop)),
            'value': round(clips_done * args.frames / elapsed, 4), 'unit': 'frames/s',
            'n_gpus': dist.get_world_size() if use_dist else 1, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 2), 'higher_is_better': True,
            'scaling': scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'arithmetic_contract': args.contract,
            'config': {'workload': '%dx%d 8-bit YUV420, coding structure %s: one %d-frame clip = %d intra-period units of %d frames '
                                   '(%d coded frames, the last %d repeat the last frame) encoded + decoded per step; synthetic '
                                   'random-init stand-in for model ms_ssim-4 (widths %s), last analysis conv calibrated so that '
                                   '%d (MOFNet) / %d (CodecNet) of the %d y feature maps are non-zero (%s operating point); %s'
                                   % (args.width, args.height, args.gop, args.frames, n_units, unit, coded, coded - args.frames, widths,
                                      active_y[0], active_y[1], widths['c_y'],
                                      'high-rate' if active_y[1] >= widths['c_y'] else 'low-rate',
                                      ('the clip sharded over %d GPUs: %d unit groups x %d ranks of temporal-layer sharding' % (world, g, r_))
                                      if sharded else ('one GPU' if world == 1 else 'one clip per GPU (replicas)')),
                       'requested_frames_per_step': args.frames, 'coded_frames_per_step': coded, 'units_per_step': n_units,
                       'nonzero_y_maps': {'mofnet': active_y[0], 'codecnet': active_y[1], 'of': widths['c_y']},
                       'parallelism': ('unit-groups x%d, level-sharded x%d%s' % (g, r_, ', levels narrower than the group in row bands'
                                                                                   if fc._banded(shard, 1, args.height, args.width) else ''))
                       if sharded else ('single GPU' if world == 1 else 'replicas x%d' % world)},
            'coded_frames_per_s': round(clips_done * coded / elapsed, 4),
            # main-stream time between events recorded after each phase was ISSUED: the decoder's entropy stage starts
            # on side streams under the encoder's last synthesis, so the two halves overlap (not comparable with the
            # host-synchronised encode_fps_rank0 / decode_fps_rank0 of rounds 1-2; `value` is wall clock and unaffected)
            'encode_main_stream_fps_rank0': round(args.steps * args.frames / stats['enc_s'], 3),
            'decode_main_stream_fps_rank0': round(args.steps * args.frames / stats['dec_s'], 3),
            'bytes_per_frame': round(stats['bytes'] / (args.steps * coded), 1),
            'closed_loop_ok': bool(closed_loop), 'stream_errors_rank0': stream_errors, 'bytes_equal_single_rank': bytes_equal,
            'row_bands': row_bands,
            # closed loop = this build's decoder on this build's bitstream: a y section written on another implementation
            # of the transforms (the reference on ATen) desynchronises at these sizes, torchac's
            # bytes are unpinned here (no wheel in the image)
            'closed_loop_scope': 'encoder and decoder of this build only',
            'parity_checked': bool(cpu and cpu.get('parity_checked')), 'quality': quality,
            'roofline': roofline, 'cpu_baseline': cpu, 'pipelined': pipelined, ('contract_v1' if args.contract == 'fp32w' else 'contract_v2'): contract_v2, 'high_rate': high_rate, 'bitstream_only_encoder': lean_encoder, 'precision_mode': precision_mode,
        }
        if other is not None:
            out['weak_scaling'] = other
        if args.tiny:
            out['invalid'] = 'tiny debug model'
        if args.widths != 'default':
            out['side_measurement'] = 'model widths %s instead of the stand-in the headline is quoted on' % widths
        if use_dist and backend != 'nccl':
            out['invalid'] = 'validation run over %s, not RCCL' % backend
        line = json.dumps(out)
    if use_dist:
        dist.barrier()  # the other ranks wait for rank 0's instrumented step before tearing RCCL down
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes to the real stdout; everything libraries print there (RCCL's version banner, its
        # warnings) was sent to stderr by main()'s redirection
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (line + '\n').encode())


if __name__ == '__main__':
    main()
