"""ArithmeticCoder: CDF build + torchac-compatible range coding + per-frame section framing
(src/real_life/bitstream.py), entirely on the GPU.

Frame bitstream = 4 sections in fixed order mofnet_z, mofnet_y, codecnet_z, codecnet_y; each is
[n_bytes 4 B BE][payload]; y payloads start with [n_nonzero_maps 1 B][map index 1 B each]; an I
frame carries two empty (4 zero bytes) MOFNet sections; an all-zero y is the single byte 0.

What differs from the reference implementation (not from its bytes): the [C,H,W,514] fp32 CDF is
never materialised -- the encoder evaluates the 2 CDF points a symbol needs, the decoder reads the
64-entry uint16 window around zero of a position's row (another kernel produced it, with sigma beside it for
the rare symbol outside), z uses a [C_z,514] table built once; there is no device->host copy of CDFs and no
temp file.
"""
import os

import numpy as np
import torch

from .. import abi, ops
from ..func_util.nn_util import get_value
from .utils import BITSTREAM_SUFFIX

SECTION_NAMES = ('mofnet_z', 'mofnet_y', 'codecnet_z', 'codecnet_y')


class PendingSection:
    """A latent of ONE frame whose symbols are known on the device but not yet range-coded.
    q / sigma are [1,h,w,c] views; `flags` is that frame's row of a batched non-zero-map tensor."""

    def __init__(self, mode, q, sigma=None, table=None, flags=None, md5=b'', batch=None):
        self.mode, self.q, self.sigma, self.table, self.flags, self.md5 = mode, q, sigma, table, flags, md5
        # (q of the whole frame batch, its sigma or None, this frame's index in it): sections of one batch get their CDF
        # bounds from ONE launch (aivc_laplace_bounds_batch / aivc_table_bounds_batch) instead of one per frame
        self.batch = batch


_MD5_LINES = None


def latent_md5(q_nhwc):
    """The 32 hex characters the reference puts in front of a section under flag_md5sum
    (src/real_life/bitstream.py:229-234): md5 of the text file np.savetxt writes for the latent flattened in
    NCHW order -- one '%.18e' line per symbol -- so a bitstream written with the flag by either code base
    verifies under the other.  Debug path: the latent comes to the host (a sync)."""
    global _MD5_LINES
    import hashlib
    if _MD5_LINES is None:
        _MD5_LINES = [('%.18e\n' % v).encode() for v in range(-abi.AC_MAX_VAL - 1, abi.AC_MAX_VAL + 1)]
    v = q_nhwc.permute(0, 3, 1, 2).to(torch.int16).cpu().numpy().astype(int).flatten()
    lines = _MD5_LINES
    off = abi.AC_MAX_VAL + 1
    return hashlib.md5(b''.join([lines[x + off] for x in v.tolist()])).hexdigest().encode()


def split_sections(frame_bytes, upto=len(SECTION_NAMES)):
    """frame bytes -> the payloads of its first `upto` sections (bytes).  The path API appends the sections of a frame to
    its file one by one and reads a section back before the later ones exist (src/real_life/bitstream.py:333-350, 352-425):
    it asks for the sections up to the one it needs; whole frames are parsed in full, a length prefix running past the
    data is a ContainerError."""
    from .cat_binary_files import _take
    out, pos = [], 0
    for name in SECTION_NAMES[:upto]:
        sec, pos = _take(frame_bytes, pos, 'section ' + name)
        out.append(sec)
    return out


_ROWS_WS = {}


def _rows_workspace(n_rows, device):
    """Caller-owned scratch for the decoder's CDF windows (64 entries = 128 B per coded symbol, + its sigma):
    grown monotonically and reused -- large allocations of varying size otherwise go back to hipMalloc
    on every batch.  One buffer per (device, stream): the entropy stage runs on its own stream.
    -> (win [n_rows, CDF_WIN] int16, sigma_pos [n_rows] float32)"""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    ws = _ROWS_WS.get(key)
    if ws is None or ws[0].shape[0] < n_rows:
        ws = None
        _ROWS_WS.pop(key, None)
        cap = int(n_rows * 1.25) + 1024
        ws = (torch.empty((cap, abi.CDF_WIN), dtype=torch.int16, device=device),
              torch.empty(cap, dtype=torch.float32, device=device))
        _ROWS_WS[key] = ws
    return ws[0][:n_rows], ws[1][:n_rows]


_PIN_POOL = {}


def _pinned(n, dtype, floor=1 << 16):
    """pinned host staging buffer from a small free-list (hipHostMalloc per level is slow); the
    buffer goes back to the pool when its EntropyJob has been collected.  `floor`: smallest buffer handed out
    (entries): the big default keeps the byte / flag buffers of a level in ONE size class, the 4-byte-per-stream
    counts of the length check ask for 64"""
    key = (dtype, max(int(floor), 1 << (max(int(n), 1) - 1).bit_length()))
    lst = _PIN_POOL.setdefault(key, [])
    t = lst.pop() if lst else torch.empty(key[1], dtype=dtype, pin_memory=True)
    return t[:n]


def _unpin(t):
    base = t._base if t._base is not None else t
    _PIN_POOL.setdefault((base.dtype, base.numel()), []).append(base)


# In-band rate check of the reference (src/real_life/bitstream.py:307-329, RESULT lines of src/real_life/encode.py:
# 153-170): with ESTIMATE_RATE on, every range-encoder stream also gets the bits its CDF bounds price it at
# (aivc_bounds_rate, one small launch per stream behind the bounds kernels, off the main stream) -- EntropyJob.est_bits.
# Off by default: logging only.
ESTIMATE_RATE = False


class EntropyJob:
    """Range-encode launches of a set of frames, possibly still running on a side stream."""

    def __init__(self, n_frames, present, heads, jobs, out_h, lens_h, offs, event, keep, est_h=None):
        self.n_frames, self.present, self.heads, self.jobs = n_frames, present, heads, jobs
        self.out_h, self.lens_h, self.offs, self.event, self.keep = out_h, lens_h, offs, event, keep
        self.est_h, self.est_bits, self.real_bytes = est_h, 0.0, 0

    def collect(self):
        """-> list of frame byte strings (waits for the side stream)."""
        payload = [[b''] * 4 for _ in range(self.n_frames)]
        if self.jobs:
            self.event.synchronize()
            out_h, lens_h = self.out_h.numpy(), self.lens_h.numpy()
            for (fi, si), (off, _cap), ln in zip(self.jobs, self.offs, lens_h):
                if int(ln) < 0:
                    raise RuntimeError('range encoder output buffer overflow')
                payload[fi][si] = out_h[off:off + int(ln)].tobytes()
            if self.est_h is not None:
                self.est_bits = float(self.est_h.numpy().sum())
                self.real_bytes = int(lens_h.sum())
                _unpin(self.est_h)
                self.est_h = None
        frames = []
        for fi in range(self.n_frames):
            blob = b''
            for si in range(4):
                if not self.present[fi][si]:
                    blob += (0).to_bytes(4, 'big')
                else:
                    body = self.heads[fi][si] + payload[fi][si]
                    blob += len(body).to_bytes(4, 'big') + body
            frames.append(blob)
        self.keep = None
        if self.jobs:
            _unpin(self.out_h)
            _unpin(self.lens_h)
            self.out_h = self.lens_h = None
        return frames


class PreparedFlags:
    """Non-zero-map flags of a set of frames on their way to the host (async D2H + event on the stream that
    produced the latents)."""

    def __init__(self, lap, flags_h, event):
        self.lap, self.flags_h, self.event = lap, flags_h, event


def prepare_finalize(frames_sections):
    """Phase 1 of launch_finalize, no host sync: queue the D2H copy of the non-zero-map flags behind the
    kernels that produced the latents and record an event.  A caller that has more GPU work to issue (the
    next dependency level) does that between prepare_finalize and launch_finalize, so the host wait of
    phase 2 overlaps with it instead of draining the main stream."""
    lap = [(fi, si, s) for fi, secs in enumerate(frames_sections) for si, s in enumerate(secs)
           if s is not None and s.mode == 'laplace']
    flags_h = None
    if lap:
        flags_d = torch.stack([s.flags for _, _, s in lap])
        flags_h = _pinned(flags_d.numel(), torch.uint8)
        flags_h.copy_(flags_d.reshape(-1), non_blocking=True)
        flags_h = (flags_h, tuple(flags_d.shape), flags_d)
    event = torch.cuda.Event()
    event.record()
    return PreparedFlags(lap, flags_h, event)


def launch_finalize(frames_sections, side_stream=None, prepared=None, fork_streams=None):
    """frames_sections: list (one entry per frame) of 4-lists of PendingSection / None.
    One host wait for all non-zero-map flags (C bytes per latent), then the CDF-bound kernels and ONE
    batched range-encode launch per 64 streams (a wavefront per stream, all concurrent) and an async
    D2H -- on `side_stream` when given, so the transforms of the next frames overlap with it; with
    `fork_streams` the 64-stream launches of a big set run side by side instead of back to back."""
    if prepared is None:
        prepared = prepare_finalize(frames_sections)
    lap = prepared.lap
    prepared.event.synchronize()  # the one host wait
    flags_h = None
    if lap:
        buf, shape, _ = prepared.flags_h
        flags_h = buf.numpy().reshape(shape).copy()
        _unpin(buf)
    present = [[s is not None for s in secs] for secs in frames_sections]
    heads = [[b''] * 4 for _ in frames_sections]
    ctx = torch.cuda.stream(side_stream) if side_stream is not None else None
    if ctx is not None:
        side_stream.wait_event(prepared.event)  # the latents of THESE frames, not whatever was issued since
        ctx.__enter__()
    try:
        jobs, bounds = [], []
        # y sections: one bounds launch per frame BATCH the sections were cut from (a dependency level's latents)
        groups = {}
        for j, (fi, si, s) in enumerate(lap):
            maps = [int(c) for c in np.nonzero(flags_h[j])[0]]
            heads[fi][si] = s.md5 + bytes([len(maps)]) + bytes(maps)
            key = id(s.batch[0]) if s.batch is not None else ('single', j)
            groups.setdefault(key, []).append((fi, si, s, maps))
        for members in groups.values():
            s0 = members[0][2]
            if s0.batch is None:
                fi, si, s, maps = members[0]
                if maps:
                    jobs.append((fi, si))
                    bounds.append(ops.laplace_bounds(s.sigma, s.q, maps))
                continue
            q_b, sigma_b, _ = s0.batch
            per_frame = [[] for _ in range(q_b.shape[0])]
            for fi, si, s, maps in members:
                per_frame[s.batch[2]] = maps
            npix = q_b.shape[1] * q_b.shape[2]
            allb, offs = ops.laplace_bounds_batch(sigma_b, q_b, per_frame)
            for fi, si, s, maps in sorted(members, key=lambda m: m[2].batch[2]):
                if maps:
                    jobs.append((fi, si))
                    bounds.append(allb[offs[s.batch[2]]:offs[s.batch[2]] + len(maps) * npix])
        # z sections likewise (every channel of every frame)
        zgroups = {}
        for fi, secs in enumerate(frames_sections):
            for si, s in enumerate(secs):
                if s is not None and s.mode == 'pmf':
                    heads[fi][si] = s.md5
                    key = id(s.batch[0]) if s.batch is not None else ('single', fi, si)
                    zgroups.setdefault(key, []).append((fi, si, s))
        for members in zgroups.values():
            s0 = members[0][2]
            if s0.batch is None:
                for fi, si, s in members:
                    jobs.append((fi, si))
                    bounds.append(ops.table_bounds(s.table, s.q))
                continue
            zb = ops.table_bounds_batch(s0.table, s0.batch[0])
            for fi, si, s in sorted(members, key=lambda m: m[2].batch[2]):
                jobs.append((fi, si))
                bounds.append(zb[s.batch[2]])
        out_h = lens_h = offs = event = est_h = None
        keep = [frames_sections, bounds]
        if jobs:
            if ESTIMATE_RATE:
                est = torch.empty(len(bounds), dtype=torch.float64, device=bounds[0].device)
                for j, b in enumerate(bounds):
                    ops.bounds_rate(b, out=est[j:j + 1])
                est_h = _pinned(est.numel(), torch.float64, floor=64)
                est_h.copy_(est, non_blocking=True)
                keep.append(est)
            out, lens, offs = ops.range_encode(bounds, streams=fork_streams)
            out_h = _pinned(out.numel(), torch.uint8)
            lens_h = _pinned(lens.numel(), torch.int32)
            out_h.copy_(out, non_blocking=True)
            lens_h.copy_(lens, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            keep += [out, lens]
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    return EntropyJob(len(frames_sections), present, heads, jobs, out_h, lens_h, offs, event, keep, est_h)


def finalize_frames(frames_sections):
    return launch_finalize(frames_sections).collect()


def finalize_frame(sections):
    return finalize_frames([sections])[0]


class ArithmeticCoder():
    def __init__(self, param):
        default = {'balle_pdf_estim_z': None, 'device': 'cpu', 'AC_MAX_VAL': abi.AC_MAX_VAL, 'flag_md5sum': False}
        self.balle_pdf_estim = get_value('balle_pdf_estim_z', param, default)
        # debug: prepend / verify the reference's feature-wise md5 on every section (bitstream.py:26-49)
        self.flag_md5sum = get_value('flag_md5sum', param, default)
        self.md5_errors = []
        # desynchronisation detector: (what, bits consumed per stream on their way to the host, event, payload lengths)
        # of every range-decode launch since the last stream_errors() call
        self.flag_check_lengths = True
        self._length_checks = []
        self._length_errors = []  # mismatches found while draining completed checks (see _watch)
        self.AC_MAX_VAL = get_value('AC_MAX_VAL', param, default)
        if self.AC_MAX_VAL != abi.AC_MAX_VAL:
            raise NotImplementedError('the kernels are built for AC_MAX_VAL = %d' % abi.AC_MAX_VAL)

    # ---- tables --------------------------------------------------------------------------------
    def z_table(self, device):
        """uint16 CDF rows [C_z][CDF_ROW] on `device` (built once, cached on the prior module)."""
        return self.balle_pdf_estim.cdf_table(device)

    @property
    def pre_computed_z_cdf(self):
        """fp32 CDF [1, C_z, 1, 1, 514] as exposed by the reference (bitstream.py:82-125)."""
        dev = next(self.balle_pdf_estim.parameters()).device
        _, cdf = self.balle_pdf_estim.cdf_table(dev, want_float=True)
        return cdf.reshape(1, -1, 1, 1, abi.LP)

    # ---- in-memory API (device tensors in NHWC, batch of frames along n) ------------------------
    def pend_z(self, q_z):
        """q_z [n,h,w,c] -> list of n PendingSection"""
        table = self.z_table(q_z.device)
        return [PendingSection('pmf', q_z[i:i + 1], table=table, md5=self._md5(q_z[i:i + 1]), batch=(q_z, None, i))
                for i in range(q_z.shape[0])]

    def pend_y(self, q_y, sigma):
        flags = ops.nonzero_flags(q_y)  # async, no sync here
        return [PendingSection('laplace', q_y[i:i + 1], sigma=sigma[i:i + 1], flags=flags[i], md5=self._md5(q_y[i:i + 1]),
                               batch=(q_y, sigma, i))
                for i in range(q_y.shape[0])]

    def _md5(self, q):
        return latent_md5(q) if self.flag_md5sum else b''

    def _strip_md5(self, payloads):
        if not self.flag_md5sum:
            return payloads, None
        return [p[32:] for p in payloads], [p[:32] for p in payloads]

    def _check_md5(self, q, sums, what):
        """bitstream.py:488-499: report, do not raise"""
        if sums is None:
            return
        for i, want in enumerate(sums):
            if latent_md5(q[i:i + 1]) != want:
                self.md5_errors.append((what, i))
                print('[Error] lossy arithmetic coding for')
                print('\t%s frame %d of the batch' % (what, i))
                print('-' * 80)

    MAX_PENDING_CHECKS = 4096

    def __getstate__(self):
        """the queue of pending length checks holds HIP events and pinned buffers: not part of a saved model
        (torch.save(model) after a decode)"""
        d = dict(self.__dict__)
        d['_length_checks'], d['_length_errors'] = [], []
        return d

    def _judge(self, entry, wait):
        """one queued check -> its mismatches (None if its decode has not finished and wait is False); the pinned
        counts go back to the pool"""
        what, host, ev, lens, _ = entry
        if wait:
            ev.synchronize()
        elif not ev.query():
            return None
        bad = []
        for i, (b, ln) in enumerate(zip(host.numpy().tolist(), lens)):
            if (int(b) + 2 + 7) // 8 != ln:
                bad.append((what, i, ln, (int(b) + 2 + 7) // 8))
        _unpin(host)
        return bad

    def _drain(self, wait=False):
        """judge the checks whose decode has completed (oldest first, stops at the first one still running unless
        `wait`): a process that never calls stream_errors() -- a service, bench.py's timed loop -- holds a handful of
        64-entry pinned buffers, not one per decode launch since it started"""
        q = self._length_checks
        k = 0
        while k < len(q):
            bad = self._judge(q[k], wait)
            if bad is None:
                break
            self._length_errors += bad
            k += 1
        del q[:k]
        if len(self._length_errors) > 65536:  # nobody reads them: keep the newest
            del self._length_errors[:32768]

    def _watch(self, what, bits, lens):
        """queue the check `len(payload) == (bits + 2 + 7) // 8` of a decode launch (include/aivc_hip.h,
        aivc_range_decode): an async copy of 4 bytes per stream + an event on the decoding stream, no host wait.
        Completed checks are judged on the way (ev.query()), so the queue stays as short as the decodes in flight."""
        if not self.flag_check_lengths or bits is None:
            return
        self._drain()
        if len(self._length_checks) >= self.MAX_PENDING_CHECKS:  # the GPU is that far behind: wait for the oldest half
            for e in self._length_checks[:self.MAX_PENDING_CHECKS // 2]:
                self._length_errors += self._judge(e, True)
            del self._length_checks[:self.MAX_PENDING_CHECKS // 2]
        host = _pinned(bits.numel(), torch.int32, floor=64)
        host.copy_(bits, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._length_checks.append((what, host, ev, list(lens), bits))

    def stream_errors(self):
        """-> [(what, index of the stream in its launch, payload bytes, bytes its decode accounts for)] of every range
        decode since the last call whose bit count contradicts its payload length, i.e. that was not decoded with the
        CDFs it was written with (another implementation's sigma, a damaged file, a wrong model): the symbols of such
        a stream are garbage from the first differing bound on, and neither torchac nor the format says so.  Waits
        for the decodes concerned."""
        self._drain(wait=True)
        bad, self._length_errors = self._length_errors, []
        return bad

    def decode_z(self, payloads, h, w, c, device):
        """list of n payloads -> q_z int16 NHWC [n,h,w,c] (pmf mode, all channels); the n streams
        are decoded concurrently."""
        payloads, sums = self._strip_md5(payloads)
        n, npix = len(payloads), h * w
        sym, bits = ops.range_decode(payloads, self.z_table(device), [0] * n, [c * npix] * n, [npix] * n, want_bits=True,
                                     flat=True)
        self._watch('z latent', bits, [len(p) for p in payloads])
        q = ops.scatter_symbols_batch(sym, [range(c)] * n, n, npix, c).view(n, h, w, c)  # one launch for the batch
        self._check_md5(q, sums, 'z latent')
        return q

    @staticmethod
    def _parse_maps(payload, c, index=0):
        """[n_maps 1 B][map index 1 B each] at the head of a y payload (src/real_life/bitstream.py:426-447) -> list of
        map indices.  The bytes come from the file: they are checked HERE, on the host, before they reach the device
        table the batched kernels index sigma with (aivc_frame_maps cannot validate a device table; the per-frame
        entry points answer AIVC_ERR_ARG to the same input, csrc/entropy.hip check_maps)."""
        from .cat_binary_files import ContainerError
        if len(payload) < 1:
            raise ContainerError('y section of frame %d of the batch is empty (no map count)' % index)
        n_maps = payload[0]
        if n_maps > c or n_maps > abi.MAX_MAPS or len(payload) < 1 + n_maps:
            raise ContainerError('y section of frame %d of the batch announces %d feature maps (latent has %d, '
                                 'section holds %d bytes)' % (index, n_maps, c, len(payload)))
        m = list(payload[1:1 + n_maps])
        if any(i >= c for i in m):
            raise ContainerError('y section of frame %d of the batch lists feature map %d of %d' % (index, max(m), c))
        return m

    def decode_y(self, payloads, sigma):
        """list of n payloads + sigma [n,h,w,c] -> q_y int16 [n,h,w,c] (zero maps restored)."""
        payloads, sums = self._strip_md5(payloads)
        n, h, w, c = sigma.shape
        npix = h * w
        maps = [self._parse_maps(p, c, i) for i, p in enumerate(payloads)]
        live = [i for i in range(n) if maps[i]]
        # ONE launch each for the batch's CDF windows and for the scatter back into [n, h, w, c] (frame f's map list and
        # stream offset come from a device table, aivc_frame_maps); the range decoder runs one wavefront per stream
        tab = ops.frame_maps_to_device(maps, npix, sigma.device)
        _, row_offs, total = tab
        sym = None
        if live:
            # 64-entry CDF windows (what the decoder's fast path reads) + sigma per position; symbols outside the
            # window make the decoder rebuild the row itself -- 128 B and 64 CDF points per symbol instead of 1040 / 514
            win, sig = _rows_workspace(total, sigma.device)
            ops.laplace_cdf_windows_batch(sigma, maps, (win, sig), table=tab)
            coded = [payloads[i][1 + len(maps[i]):] for i in live]
            sym, bits = ops.range_decode(coded, win, [row_offs[i] for i in live], [len(maps[i]) * npix for i in live],
                                         [0] * len(live), sigma_pos=sig, want_bits=True, flat=True)
            self._watch('y latent', bits, [len(p) for p in coded])
        if sym is None:
            sym = torch.zeros(1, dtype=torch.int16, device=sigma.device)
        q = ops.scatter_symbols_batch(sym, maps, n, npix, c, table=tab).view(n, h, w, c)
        self._check_md5(q, sums, 'y latent')
        return q

    # ---- path-based API with the reference's signatures (NCHW float tensors, one file per frame) --
    def encode(self, param):
        default = {'x': None, 'mode': 'laplace', 'sigma': None, 'bitstream_path': None, 'flag_debug': True,
                   'latent_name': '', 'flag_md5sum': False}
        x = get_value('x', param, default)
        mode = get_value('mode', param, default)
        sigma = get_value('sigma', param, default)
        path = get_value('bitstream_path', param, default)
        latent_name = get_value('latent_name', param, default)
        if not path.endswith(BITSTREAM_SUFFIX):
            path += BITSTREAM_SUFFIX
        # torchac.encode_float_cdf(..., check_input_bounds=True) raises on symbols outside the alphabet 0 .. Lp - 2 = 512
        # (src/real_life/bitstream.py:280-281), i.e. on values outside [-AC_MAX_VAL, AC_MAX_VAL]; one device sync
        if x.numel():
            lo, hi = (float(v) for v in torch.aminmax(x))
            if lo < -self.AC_MAX_VAL or hi > self.AC_MAX_VAL:
                raise ValueError('ArithmeticCoder.encode: values in [%g, %g] outside [-%d, %d]'
                                 % (lo, hi, self.AC_MAX_VAL, self.AC_MAX_VAL))
        q = ops.to_nhwc(x).to(torch.int16)
        keep, self.flag_md5sum = self.flag_md5sum, bool(get_value('flag_md5sum', param, default))
        try:
            sec = (self.pend_y(q, ops.to_nhwc(sigma)) if mode == 'laplace' else self.pend_z(q))[0]
        finally:
            self.flag_md5sum = keep
        slots = [None] * 4
        slots[SECTION_NAMES.index(latent_name)] = sec
        body = split_sections(finalize_frame(slots))[SECTION_NAMES.index(latent_name)]
        blob = len(body).to_bytes(4, 'big') + body
        if latent_name == 'codecnet_z' and not os.path.isfile(path):
            blob = (0).to_bytes(8, 'big') + blob
        with open(path, 'ab') as f:
            f.write(blob)
        if get_value('flag_debug', param, default):
            self._debug_report(x, sigma, mode, body, len(blob), path, latent_name, param)

    def _debug_report(self, x, sigma, mode, body, nbytes_written, path, latent_name, param):
        """flag_debug of the reference (src/real_life/bitstream.py:306-350): estimated against real rate, then
        a decode of what was just written.  Debug path: the rate-estimation kernels (csrc/rate.hip), one sync."""
        md5 = 32 if param.get('flag_md5sum', False) else 0
        if mode == 'laplace':
            nb_sent = body[md5]
            header_overhead = md5 + 1 + nb_sent
            proba = ops.laplace_prob(x.float(), None, sigma.expand_as(x))
        else:
            nb_sent = x.shape[1]
            header_overhead = md5
            proba = self.balle_pdf_estim(x.float())
        _, bits = ops.rate_bits(proba, 2 ** -16, 1.)
        estimated_rate = float(bits.cpu()) / 8000 + 1e-3
        real_rate = (len(body) + 4) / 1000
        print('Arithmetic coding of      : ' + str(path.split('/')[-1].rstrip(BITSTREAM_SUFFIX)) + ' ' + latent_name)
        print('Number of ft. maps sent   : ' + str(nb_sent))
        print('Bitrate estimation [kByte]: ' + '%.3f' % estimated_rate)
        print('Real bitstream     [kByte]: ' + '%.3f' % real_rate)
        print('Rate overhead          [%]: ' + '%.1f' % ((real_rate / estimated_rate - 1) * 100))
        print('Absolute overhead  [Kbyte]: ' + '%.3f' % (real_rate - estimated_rate))
        print('Header overhead     [byte]: ' + '%.1f' % header_overhead)
        print('Nb. bytes in file   [byte]: ' + '%.1f' % nbytes_written)
        x_decoded = self.decode(dict(param, data_dim=x.size(), device=x.device, flag_debug=False))
        if torch.all(torch.eq(x, x_decoded.to(x.dtype))):
            print('Ok! Entropy coding is lossless\n')
        else:
            print('-' * 80)
            print('Ko! Entropy coding is not lossless: ' + str((x_decoded - x).abs().sum()) + '\n')
            print('-' * 80)

    def decode(self, param):
        default = {'mode': 'laplace', 'sigma': None, 'bitstream_path': None, 'data_dim': None, 'device': 'cpu',
                   'flag_debug': True, 'latent_name': '', 'flag_md5sum': False}
        mode = get_value('mode', param, default)
        sigma = get_value('sigma', param, default)
        path = get_value('bitstream_path', param, default)
        data_dim = get_value('data_dim', param, default)
        device = get_value('device', param, default)
        latent_name = get_value('latent_name', param, default)
        if not path.endswith(BITSTREAM_SUFFIX):
            path += BITSTREAM_SUFFIX
        with open(path, 'rb') as f:
            payload = split_sections(f.read(), SECTION_NAMES.index(latent_name) + 1)[-1]
        keep, self.flag_md5sum = self.flag_md5sum, bool(get_value('flag_md5sum', param, default))
        try:
            n_err = len(self.md5_errors)
            if mode == 'laplace':
                q = self.decode_y([payload], ops.to_nhwc(sigma))
            else:
                b, c, h, w = data_dim
                q = self.decode_z([payload], h, w, c, torch.device(device))
            if self.flag_md5sum and len(self.md5_errors) == n_err:
                print('All good for ' + path + ' ' + latent_name)
        finally:
            self.flag_md5sum = keep
        return ops.to_nchw_view(q.float())
