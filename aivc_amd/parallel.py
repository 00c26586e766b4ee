"""Multi-GPU: one process per GPU, intra-period units sharded across ranks.

Units are independent (each starts with its own I frame, references never cross a unit boundary,
src/real_life/decode.py:239), so unit u is coded by rank u % world and the container is assembled
on rank 0 from the gathered GOP records: the bytes are identical to a single-GPU run by
construction.  The only collective on the data path is the gather of a few kB..MB of bitstream
(and, for decode, of the 8-bit frames); weights are broadcast once at start-up."""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def rank_world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


# ---- start-up and a watchdog for the first contact with RCCL -------------------------------------------------------
_TRAIL = []        # the last collectives this rank issued: (name, wall time)
_WATCH = {'thread': None, 'waiting': None}


def init_process_group(device=None, backend=None, timeout_s=None):
    """torch.distributed start-up for one process per GPU: RCCL ('nccl') bound to `device` (eager communicator
    creation; sub-groups are split from it), gloo where AIVC_DIST_BACKEND says so (the one-GPU test boxes).  A finite
    timeout (AIVC_DIST_TIMEOUT_S, default 300 s) so that a rank that never arrives ends the job with PyTorch's
    collective dump instead of hanging the launcher, and a watchdog thread that names what this rank is waiting in."""
    import datetime
    import os
    if dist.is_initialized():
        return
    backend = backend or os.environ.get('AIVC_DIST_BACKEND', 'nccl')
    timeout = datetime.timedelta(seconds=float(timeout_s or os.environ.get('AIVC_DIST_TIMEOUT_S', '300')))
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=device, timeout=timeout)
    else:
        dist.init_process_group(backend, timeout=timeout)
    _start_watchdog()


def _note(name):
    """record a collective about to be issued (cheap: a list append)"""
    import time
    _TRAIL.append((name, time.time()))
    if len(_TRAIL) > 32:
        del _TRAIL[:16]


class _host_wait:
    """`with _host_wait('what'):` around a point where the HOST blocks on communication (a .cpu() of a gathered tensor,
    a barrier, a gloo collective): if it lasts longer than AIVC_DIST_WARN_S (60 s) the watchdog prints, once, which wait
    it is and the collectives this rank issued before it -- under RCCL the enqueue of a collective returns at once and a
    missing peer only shows at the next host synchronisation, far from its cause."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        import time
        _WATCH['waiting'] = (self.name, time.time(), False)

    def __exit__(self, *exc):
        _WATCH['waiting'] = None
        return False


def _start_watchdog():
    import os
    import sys
    import threading
    import time
    if _WATCH['thread'] is not None:
        return
    warn = float(os.environ.get('AIVC_DIST_WARN_S', '60'))

    def run():
        while True:
            time.sleep(min(5.0, warn / 4))
            w = _WATCH['waiting']
            if w is not None and not w[2] and time.time() - w[1] > warn:
                _WATCH['waiting'] = (w[0], w[1], True)
                trail = ', '.join('%s (%.0f s ago)' % (n, time.time() - t) for n, t in _TRAIL[-6:])
                sys.stderr.write('[aivc_amd.parallel] rank %d has been waiting %.0f s in "%s"; last collectives issued: %s\n'
                                 % (dist.get_rank() if dist.is_initialized() else -1, time.time() - w[1], w[0], trail or 'none'))
                sys.stderr.flush()
    t = threading.Thread(target=run, name='aivc-dist-watchdog', daemon=True)
    t.start()
    _WATCH['thread'] = t


def broadcast_model(model, src=0):
    """ONE broadcast of the weights from `src` (RCCL over xGMI on GPUs, gloo on CPU): every parameter and buffer of a
    dtype travels in one flat tensor (a real model has ~370 of them; fp32 except the odd integer buffer)."""
    if not is_dist():
        return model
    with torch.no_grad():
        tensors = [t.data for t in list(model.parameters()) + list(model.buffers())]
        by_type = {}
        for t in tensors:
            by_type.setdefault((t.dtype, t.device), []).append(t)
        for (dtype, device), ts in by_type.items():
            flat = torch.cat([t.reshape(-1) for t in ts])
            _note('broadcast weights %s x%d' % (dtype, flat.numel()))
            dist.broadcast(flat, src)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
    # writing through .data does not bump Parameter._version, which is what the packed-weight / GDN / z-table
    # cache is stamped with: anything cached before the broadcast would stay stale on the receiving ranks
    from .layers import _cache
    _cache.clear()
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        with _host_wait('device sync after the weight broadcast'):
            torch.cuda.synchronize()  # the codec's side streams read the parameters without waiting for this stream
    return model


def unit_owner(u, world):
    return u % world


def gather_gops(local_gops, dst=0, device=None):
    """local_gops: list over ALL units with None for units coded elsewhere.  Returns the complete
    list on rank `dst` (None on the others).  Two tensor collectives (lengths, padded payload): no pickling."""
    rank, world = rank_world()
    if world == 1:
        return local_gops
    keys = list(range(len(local_gops)))
    allb = gather_bytes_all({u: g for u, g in enumerate(local_gops) if g is not None}, keys, None, world, device)
    if rank != dst:
        return None
    assert all(u in allb for u in keys), 'a unit was coded by no rank'
    return [allb[u] for u in keys]


def encode_video_sharded(frame_codec, frames, gop_name, idx_starting_frame=0, idx_rate=0., return_enc=False):
    """Every rank passes the same `frames`; returns the full bitstream on rank 0 (None elsewhere); with return_enc also
    this rank's encode_video record (its units' reconstructions, None for the others')."""
    rank, world = rank_world()
    enc = frame_codec.encode_video(frames, gop_name, idx_starting_frame, idx_rate=idx_rate,
                                   unit_filter=lambda u: unit_owner(u, world) == rank)
    dev = getattr(frames[0]['y'], 'device', None)
    dev = dev if isinstance(dev, torch.device) else None
    gops = gather_gops(enc['gops'], device=dev)
    data_dim = enc['data_dim']
    if world > 1:  # ranks without a unit do not know the latent sizes: element-wise max of six int64
        cdev = _comm_device(None, dev)
        v = [-1] * 6 if data_dim is None else [*data_dim['x'], *data_dim['y'], *data_dim['z']]
        t = torch.tensor(v, dtype=torch.int64, device=cdev)
        _note('all_reduce latent sizes')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        with _host_wait('latent sizes to the host'):
            v = [int(x) for x in t.cpu()]
        data_dim = {'x': (v[0], v[1]), 'y': (v[2], v[3]), 'z': (v[4], v[5]), 'x_uv': ((v[0] + 1) // 2, (v[1] + 1) // 2)}
    if gops is None:
        return (None, enc) if return_enc else None
    blob = frame_codec.assemble_video(dict(enc, gops=gops, data_dim=data_dim))
    return (blob, enc) if return_enc else blob


def decode_video_sharded(frame_codec, blob, device=None):
    """Every rank holds the bitstream; each decodes its units; rank 0 returns all frames as lists of
    dicts of CPU uint8 tensors (None elsewhere)."""
    rank, world = rank_world()
    frames, data_dim, first, last = frame_codec.decode_video(
        blob, device, unit_filter=lambda u: unit_owner(u, world) == rank)
    if world == 1:
        return [None if f is None else {k: f[k].cpu() for k in 'yuv'} for f in frames]
    # one all_gather of the 8-bit planes (every rank sends `per` frame slots of h*w + 2*hc*wc bytes, its decoded
    # frames first): a tensor collective on the device under RCCL, no pickling
    h, w = data_dim['x']
    hc, wc = (h + 1) // 2, (w + 1) // 2
    fsz = h * w + 2 * hc * wc
    owner = [None] * len(frames)  # (rank, slot) of every frame: units are dealt round-robin, known everywhere
    counts = [0] * world
    # every unit's OWN length: a container may mix coding structures (decode_video / decode_units accept that), and
    # the padded frames of the last unit are cut from the end of `frames`
    i = 0
    for u, n_u in enumerate(unit_lengths(blob)):
        for _ in range(n_u):
            if i < len(frames):
                owner[i] = (unit_owner(u, world), counts[unit_owner(u, world)])
                counts[unit_owner(u, world)] += 1
                i += 1
    assert i == len(frames), 'the container holds %d frames, decode_video returned %d' % (i, len(frames))
    per = max(counts)
    dev = next(f['y'].device for f in frames if f is not None) if any(f is not None for f in frames) else device
    cdev = _comm_device(None, dev)
    send = torch.zeros((per, fsz), dtype=torch.uint8, device=cdev)
    mine = [f for f in frames if f is not None]
    if mine:
        torch.cat([f[k].reshape(-1).to(cdev) for f in mine for k in 'yuv'], out=send.view(-1)[:len(mine) * fsz])
    recv = torch.empty((world * per, fsz), dtype=torch.uint8, device=cdev)
    _note('all_gather decoded planes')
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    if rank != 0:
        return None
    with _host_wait('decoded planes to the host'):
        recv = recv.cpu()
    out = []
    for r, slot in owner:
        row = recv[r * per + slot]
        out.append({'y': row[:h * w].view(1, h, w), 'u': row[h * w:h * w + hc * wc].view(1, hc, wc),
                    'v': row[h * w + hc * wc:].view(1, hc, wc)})
    return out


def unit_lengths(blob):
    """number of frames of every intra-period unit of the video `blob`, in order"""
    from .real_life import cat_binary_files as container
    _, _, _, gops = container.unpack_video(blob)
    return [len(container.unpack_gop(g)[2]) for g in gops]


# ---- one clip over all GPUs: unit groups x temporal-layer sharding ---------------------------
# BASELINE configs[3] is ONE 128-frame 1080p clip on 8 GPUs: 4 intra-period units.  Units go to
# G = min(world, n_units) groups of R = world // G ranks; inside a group the frames of one dependency level
# (they only depend on earlier levels) are dealt round-robin to its ranks.  The only exchanges are
#   * per level: one all_gather of the new 8-bit reconstructions inside the group (3.1 MB per 1080p frame over
#     xGMI) -- they are the references of the next levels;
#   * once per clip: the frame bitstreams inside the group, then the GOP records to every rank
# all as TENSOR collectives (uint8 payload + int64 lengths): no pickling, no host round trip on RCCL.
# Output bytes are identical to a single process by construction (tests: gloo on CPU with the oracle as the frame
# coder, gloo + two processes on one GPU with the HIP codec).
def _comm_device(group=None, device=None):
    """tensors handed to collectives: CUDA under RCCL ('nccl'), host memory under gloo"""
    if dist.get_backend(group) == 'nccl':
        return device if device is not None and device.type == 'cuda' else torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def gather_bytes_all(mine, keys, group, n_ranks, device=None):
    """mine: {key: bytes} held by this rank; keys: ordered list of all keys (same on all ranks).
    -> {key: bytes} complete on every rank of `group` (None: all ranks).  Two tensor collectives: lengths, padded
    payload -- uint8 / int64 tensors on the device under RCCL, host memory under gloo; no pickling."""
    cdev = _comm_device(group, device)
    lens = torch.tensor([len(mine[k]) if k in mine else -1 for k in keys], dtype=torch.int64, device=cdev)
    all_lens = torch.empty((n_ranks, len(keys)), dtype=torch.int64, device=cdev)
    _note('all_gather byte lengths')
    dist.all_gather_into_tensor(all_lens.view(-1), lens, group=group)
    with _host_wait('byte lengths to the host'):
        all_lens = all_lens.cpu()
    totals = all_lens.clamp_min(0).sum(dim=1)
    cap = max(int(totals.max()), 1)
    payload = bytearray()
    for k in keys:
        if k in mine:
            payload += mine[k]
    buf = torch.zeros(cap, dtype=torch.uint8)
    if payload:
        buf[:len(payload)] = torch.frombuffer(payload, dtype=torch.uint8)
    buf = buf.to(cdev)
    recv = torch.empty((n_ranks, cap), dtype=torch.uint8, device=cdev)
    _note('all_gather byte payload')
    dist.all_gather_into_tensor(recv.view(-1), buf, group=group)
    with _host_wait('byte payload to the host'):
        recv = recv.cpu().numpy()
    out = {}
    for r in range(n_ranks):
        pos = 0
        for j, k in enumerate(keys):
            n = int(all_lens[r, j])
            if n >= 0:
                out.setdefault(k, recv[r, pos:pos + n].tobytes())
                pos += n
    return out


_SHARDS = {}


def clip_shard(n_units, device=None):
    """The ClipShard of (n_units, world, device), built once per process: every construction calls dist.new_group
    per unit group, which under RCCL allocates a communicator that is never freed -- a service coding many clips
    must not build one per clip.  COLLECTIVE on first use for a key: every rank has to make the same first call."""
    _, world = rank_world()
    key = (int(n_units), world, None if device is None else str(device), dist.get_backend() if is_dist() else None)
    sh = _SHARDS.get(key)
    if sh is None:
        sh = _SHARDS[key] = ClipShard(n_units, device)
    return sh


class ClipShard:
    """Partition of one clip's work over the ranks.  Construction is a COLLECTIVE call: every rank must construct
    it (sub-groups are created with dist.new_group in the same order everywhere); prefer clip_shard(), which builds
    one per (n_units, world, device) and keeps it."""

    def __init__(self, n_units, device=None):
        self.rank, self.world = rank_world()
        self.G = max(1, min(self.world, n_units))
        self.R = max(1, self.world // self.G)
        active = self.G * self.R  # ranks beyond (world not a multiple of G) idle
        self.active = self.rank < active
        self.group_id = self.rank // self.R if self.active else None
        self.local = self.rank % self.R if self.active else 0
        self.units = [u for u in range(n_units) if self.active and u % self.G == self.group_id]
        self.n_units = n_units
        self.device = device
        self.pg = None
        self.backend = dist.get_backend() if is_dist() else None
        self.band_levels = None  # row bands for levels narrower than the group: None = FrameCodec._banded's rule
        if self.world > 1:
            for g in range(self.G):
                ranks = list(range(g * self.R, (g + 1) * self.R))
                # every rank calls new_group for EVERY group, in the same order (a collective over the world)
                _note('new_group %s' % ranks)
                pg = dist.new_group(ranks) if self.R > 1 else None
                if g == self.group_id:
                    self.pg = pg
        self.leaders = [g * self.R for g in range(self.G)]
        self._bufs = {}  # persistent exchange buffers, (slots, frame bytes, device) -> (send, recv)

    # ---- who codes what ---------------------------------------------------------------------------------
    def mine(self, items):
        return items[self.local::self.R]

    def bands(self):
        """row-band context of this rank's group (aivc_amd/bands.py): levels with fewer frames than the group has
        ranks are coded one frame at a time, every rank a band of rows (FrameCodec._banded)"""
        if getattr(self, '_bands', None) is None:
            from .bands import BandCtx, DistComm
            ranks = list(range(self.group_id * self.R, (self.group_id + 1) * self.R))
            dev = self.device if self.device is not None else torch.device('cuda', torch.cuda.current_device())
            self._bands = BandCtx(DistComm(self.pg, ranks, self.local), dev)
        return self._bands

    # ---- exchanges inside the group ------------------------------------------------------------------------
    def exchange_frames(self, items, my_recs, h, w, device):
        """items: every frame of a level (same order on all ranks of the group); my_recs: reconstructions of
        self.mine(items).  -> reconstructions of all items (uint8 plane dicts on `device`)."""
        if self.R == 1:
            return my_recs
        cdev = _comm_device(self.pg, device)
        per = (len(items) + self.R - 1) // self.R  # every rank sends `per` frame slots (unused ones: stale bytes)
        hc, wc = (h + 1) // 2, (w + 1) // 2
        fsz = h * w + 2 * hc * wc
        # the send side is persistent (one buffer per shape); the receive side is allocated per call: the frames
        # handed out below are views of it and live as references for the rest of the unit
        key = (per, fsz, str(cdev))
        send = self._bufs.get(key)
        if send is None:
            send = self._bufs[key] = torch.zeros((per, fsz), dtype=torch.uint8, device=cdev)
        if my_recs:  # ONE launch packs every plane of this rank's frames (was a cat + copy per frame)
            flat = [r[k].reshape(-1) if r[k].device == cdev else r[k].reshape(-1).to(cdev) for r in my_recs for k in 'yuv']
            torch.cat(flat, out=send.view(-1)[:len(my_recs) * fsz])
        recv = torch.empty((self.R * per, fsz), dtype=torch.uint8, device=cdev)
        # asynchronous: under RCCL the gather runs on the communicator's own stream (it waits for the packing on
        # the current stream, nothing else does: the entropy coder's side streams keep running); the current stream
        # joins it at work.wait() -- the frames are the references of the very next level, so that is right away
        _note('all_gather level reconstructions')
        work = dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.pg, async_op=True)  # flat: gloo wants 1-D
        with _host_wait('level reconstructions'):
            work.wait()
        recv = recv.to(device)
        out = [None] * len(items)
        for j in range(len(items)):
            row = recv[(j % self.R) * per + j // self.R]
            out[j] = {'y': row[:h * w].view(1, h, w), 'u': row[h * w:h * w + hc * wc].view(1, hc, wc),
                      'v': row[h * w + hc * wc:].view(1, hc, wc)}
        return out

    def _gather_bytes(self, mine, keys, group, n_ranks, device=None):
        return gather_bytes_all(mine, keys, group, n_ranks, device or self.device)
    def gather_bytes(self, mine, keys):
        """frame bitstreams inside the group"""
        if self.R == 1:
            return dict(mine)
        return self._gather_bytes(mine, keys, self.pg, self.R)

    def agree(self, data_dim):
        """latent sizes are known to the ranks that coded a frame; every rank of the group returns them"""
        if self.R == 1:
            return data_dim
        cdev = _comm_device(self.pg, self.device)
        v = [-1] * 6 if data_dim is None else [*data_dim['x'], *data_dim['y'], *data_dim['z']]
        t = torch.tensor(v, dtype=torch.int64, device=cdev)
        allv = torch.empty((self.R, 6), dtype=torch.int64, device=cdev)
        _note('all_gather latent sizes (group)')
        dist.all_gather_into_tensor(allv.view(-1), t, group=self.pg)
        with _host_wait('latent sizes (group) to the host'):
            v = [int(x) for x in allv.cpu().max(dim=0).values]
        return {'x': (v[0], v[1]), 'y': (v[2], v[3]), 'z': (v[4], v[5]), 'x_uv': ((v[0] + 1) // 2, (v[1] + 1) // 2)}

    # ---- across the groups ------------------------------------------------------------------------------------
    def gather_units(self, blobs_by_unit, data_dim=None):
        """blobs_by_unit: {unit: GOP record} of this rank's group.  -> ([record per unit], data_dim) on EVERY rank."""
        if self.world == 1:
            return [blobs_by_unit[u] for u in range(self.n_units)], data_dim
        keys = list(range(self.n_units)) + ['dd']
        mine = {}
        if self.active and self.local == 0:  # one contributor per group
            mine = dict(blobs_by_unit)
            if data_dim is not None and self.group_id == 0:
                mine['dd'] = b''.join(int(x).to_bytes(4, 'big') for x in (*data_dim['x'], *data_dim['y'], *data_dim['z']))
        allb = self._gather_bytes(mine, keys, None, self.world)
        dd = None
        if 'dd' in allb:
            v = [int.from_bytes(allb['dd'][i:i + 4], 'big') for i in range(0, 24, 4)]
            dd = {'x': (v[0], v[1]), 'y': (v[2], v[3]), 'z': (v[4], v[5]), 'x_uv': ((v[0] + 1) // 2, (v[1] + 1) // 2)}
        return [allb[u] for u in range(self.n_units)], dd


def encode_clip(frame_codec, units, gop_name, idx_rate=0., shard=None):
    """Strong scaling: ONE clip (list of intra-period units, every rank passes the same frames) over all ranks.
    -> ([GOP record per unit], data_dim) on every rank; bytes identical to frame_codec.encode_units(units)."""
    shard = shard or clip_shard(len(units), units[0][0]['y'].device)
    mine = {}
    data_dim = None
    if shard.units:
        blobs, _, data_dim = frame_codec.encode_units([units[u] for u in shard.units], gop_name, idx_rate, shard=shard)
        mine = dict(zip(shard.units, blobs))
    return shard.gather_units(mine, data_dim)


def decode_clip(frame_codec, gop_blobs, data_dim, device=None, shard=None):
    """-> {unit: [reconstructions in display order]} for the units of this rank's group (the frames stay on the
    GPUs that decoded them; every rank of a group holds all frames of the group's units)."""
    shard = shard or clip_shard(len(gop_blobs), device)
    if not shard.units:
        return {}
    recs = frame_codec.decode_units([gop_blobs[u] for u in shard.units], data_dim, device, shard=shard)
    return dict(zip(shard.units, recs))


# ---- generic level-by-level drivers (any frame coder with encode_batch / decode_batch: the CPU tests run them
# with the oracle behind that interface; the HIP codec has the same logic inside FrameCodec.encode_units /
# decode_units, where it keeps the single-GPU stream scheduling) ----------------------------------------------------
def encode_units_level_sharded(frame_codec, units, gop_name, idx_rate=0., shard=None):
    """Every rank passes the same `units` (those of its group) and returns the same (gop blobs, data_dim)."""
    from .codec import frame_index
    from .func_util.GOP_structure import coding_levels, generate_gop_struct
    from .real_life import cat_binary_files as container
    from .real_life import header as hdr
    from .real_life.bitstream import finalize_frames
    shard = shard or _whole_world_shard(units[0][0]['y'].device)
    gop = generate_gop_struct(gop_name)
    names = sorted(gop, key=frame_index)
    rec = [dict() for _ in units]
    fbytes = {}
    data_dim = None
    h, w = units[0][0]['y'].shape[-2:]
    dev = units[0][0]['y'].device
    for level in coding_levels(gop):
        for ftype in sorted({gop[f]['type'] for f in level}):
            items = [(u, f) for u in range(len(units)) for f in level if gop[f]['type'] == ftype]
            mine = shard.mine(items)
            my_recs = []
            for s in range(0, len(mine), frame_codec.max_batch):
                chunk = mine[s:s + frame_codec.max_batch]
                out = frame_codec.encode_batch([units[u][frame_index(f)] for u, f in chunk],
                                               [rec[u].get(gop[f]['prev_ref']) for u, f in chunk],
                                               [rec[u].get(gop[f]['next_ref']) for u, f in chunk], ftype, idx_rate)
                data_dim = out['data_dim']
                for it, b in zip(chunk, finalize_frames(out['sections'])):
                    fbytes[it] = b
                my_recs += out['rec']
            for (u, f), rc in zip(items, shard.exchange_frames(items, my_recs, h, w, dev)):
                rec[u][f] = rc
    keys = [(u, f) for u in range(len(units)) for f in names]
    fbytes = shard.gather_bytes(fbytes, keys)
    data_dim = shard.agree(data_dim)
    head = hdr.gop_header_bytes(gop_name, idx_rate)
    blobs = [container.pack_gop(head, [fbytes[(u, f)] for f in names]) for u in range(len(units))]
    return blobs, data_dim


def decode_units_level_sharded(frame_codec, gop_blobs, data_dim, device=None, shard=None):
    """Like FrameCodec.decode_units for ONE GOP structure, frames of every dependency level round-robin over
    the ranks of the group; every rank returns the same reconstructions."""
    from .codec import frame_index
    from .func_util.GOP_structure import coding_levels, generate_gop_struct
    from .real_life import cat_binary_files as container
    device = device if device is not None else (torch.device('cuda', torch.cuda.current_device())
                                                if torch.cuda.is_available() else torch.device('cpu'))
    shard = shard or _whole_world_shard(device)
    parsed = [container.unpack_gop(g) for g in gop_blobs]
    gop_name, idx_rate = parsed[0][0], parsed[0][1]
    if any((p[0], p[1]) != (gop_name, idx_rate) for p in parsed):
        raise ValueError('decode_units_level_sharded: all units must share one GOP structure and rate index')
    gop = generate_gop_struct(gop_name)
    names = sorted(gop, key=frame_index)
    h, w = data_dim['x']
    rec = [dict() for _ in gop_blobs]
    for level in coding_levels(gop):
        for ftype in sorted({gop[f]['type'] for f in level}):
            items = [(u, f) for u in range(len(gop_blobs)) for f in level if gop[f]['type'] == ftype]
            mine = shard.mine(items)
            my_recs = []
            for s in range(0, len(mine), frame_codec.max_batch):
                chunk = mine[s:s + frame_codec.max_batch]
                my_recs += frame_codec.decode_batch([parsed[u][2][frame_index(f)] for u, f in chunk],
                                                    [rec[u].get(gop[f]['prev_ref']) for u, f in chunk],
                                                    [rec[u].get(gop[f]['next_ref']) for u, f in chunk], ftype, data_dim,
                                                    idx_rate, device)
            for (u, f), rc in zip(items, shard.exchange_frames(items, my_recs, h, w, device)):
                rec[u][f] = rc
    return [[rec[u][f] for f in names] for u in range(len(gop_blobs))]


def _whole_world_shard(device):
    """all ranks form ONE group (level sharding only): ClipShard of a single unit"""
    return clip_shard(1, device)
