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


def broadcast_model(model, src=0):
    """One broadcast of every parameter/buffer from `src` (RCCL over xGMI on GPUs, gloo on CPU)."""
    if not is_dist():
        return model
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src)
    return model


def unit_owner(u, world):
    return u % world


def gather_gops(local_gops, dst=0):
    """local_gops: list over ALL units with None for units coded elsewhere.  Returns the complete
    list on rank `dst` (None on the others)."""
    rank, world = rank_world()
    if world == 1:
        return local_gops
    gathered = [None] * world if rank == dst else None
    dist.gather_object(local_gops, gathered, dst=dst)
    if rank != dst:
        return None
    out = list(local_gops)
    for r, lst in enumerate(gathered):
        for u, g in enumerate(lst):
            if g is not None:
                out[u] = g
    assert all(g is not None for g in out), 'a unit was coded by no rank'
    return out


def encode_video_sharded(frame_codec, frames, gop_name, idx_starting_frame=0, idx_rate=0.):
    """Every rank passes the same `frames`; returns the full bitstream on rank 0 (None elsewhere)."""
    rank, world = rank_world()
    enc = frame_codec.encode_video(frames, gop_name, idx_starting_frame, idx_rate=idx_rate,
                                   unit_filter=lambda u: unit_owner(u, world) == rank)
    gops = gather_gops(enc['gops'])
    dims = [enc['data_dim']]
    if world > 1:
        all_dims = [None] * world
        dist.all_gather_object(all_dims, enc['data_dim'])
        dims = [d for d in all_dims if d is not None]
    if gops is None:
        return None
    enc = dict(enc, gops=gops, data_dim=dims[0])
    return frame_codec.assemble_video(enc)


def decode_video_sharded(frame_codec, blob, device=None):
    """Every rank holds the bitstream; each decodes its units; rank 0 returns all frames as lists of
    dicts of CPU uint8 tensors (None elsewhere)."""
    rank, world = rank_world()
    frames, data_dim, first, last = frame_codec.decode_video(
        blob, device, unit_filter=lambda u: unit_owner(u, world) == rank)
    local = [None if f is None else {k: f[k].cpu() for k in 'yuv'} for f in frames]
    if world == 1:
        return local
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    if rank != 0:
        return None
    out = list(local)
    for lst in gathered:
        for i, f in enumerate(lst):
            if f is not None:
                out[i] = f
    return out
