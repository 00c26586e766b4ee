"""Row bands: ONE frame's transforms spread over the R ranks of a unit group (SURVEY.md 8e "spatial halo tiling",
BASELINE configs[4]: a single 4K intra-period unit on 8 GPUs).

The dependency levels of a hierarchical GOP are 1, 1, 1, 2, 4, ... frames wide; a level narrower than the group leaves
ranks idle under frame sharding (parallel.ClipShard).  Here every rank computes a horizontal BAND of each feature map
of such a frame and the ranks exchange the few boundary rows the next layer's kernel window reaches into -- layer by
layer (recomputing halos instead would cost ~5x: the receptive field of one conditional coder is ~550 rows).

Bit-exactness is by construction: every output element of a conv is one fixed-order fmaf chain over its own window
(include/aivc_hip.h), whatever tensor the window is cut from.  A rank runs the UNCHANGED kernels on a slab = its band
plus halo rows:

  * replicate padding (src/layers/misc/custom_conv_layers.py:145-153) is the kernel's clamp at the edge of the tensor
    it is given: a slab starts / ends at a true image edge exactly where the band does, and inside the image every row
    a valid output needs is present, so the clamp is only ever hit where the reference pads;
  * stride-2 convs need the slab's first row on an even frame row (local output j = frame output j + a / 2);
    transposed convs (:206-223) zero-extend beyond the tensor, which is what the format does at the true edges only --
    inside the image the rows a valid output needs are, again, present;
  * rows of the slab's output that were computed from missing / clamped-too-early rows are NOT valid: a Band records
    which rows are, and nothing downstream reads the others.

Partition: the rows of the coarsest grid (the y latent) are split evenly; the boundaries at finer levels are those
times 2^k (clamped to the tensor's height), so stride-2 / transposed layers map bands onto bands and only the kernel
window's overhang (1-2 rows per side) travels.  The hyperprior (1/16 .. 1/64 resolution, tiny) and the entropy coder
(serial by format) are NOT split: y and the shortcut latent are all-gathered and every rank runs them redundantly --
deterministic arithmetic makes the results identical everywhere; the group's first rank keeps the bitstream sections.
"""
import threading

import torch

from . import abi, ops


# ---- geometry -----------------------------------------------------------------------------------------------------
def conv_out_rows(mode, h, k, stride, pad):
    if mode == abi.MODE_CONV:
        return (h + 2 * pad - k) // stride + 1
    return 2 * h if mode == abi.MODE_TCONV else h


def need_rows(mode, k, stride, pad, o0, o1, h_in):
    """input rows [lo, hi) that outputs [o0, o1) read, clipped to the tensor (beyond it: replicate padding for a conv,
    zeros for a transposed conv -- both are what the kernels do at the edge of whatever they are given)"""
    if o1 <= o0:
        return 0, 0
    if mode == abi.MODE_CONV:
        lo, hi = stride * o0 - pad, stride * (o1 - 1) - pad + k
    elif mode == abi.MODE_TCONV:  # output o takes input i = (o + tp - ky) / 2 for the ky of its parity, tp = (k + 1) / 2 - 1
        tp = (k + 1) // 2 - 1
        lo, hi = -((-(o0 + tp - (k - 1))) // 2), (o1 - 1 + tp) // 2 + 1
    else:
        lo, hi = o0, o1
    lo, hi = max(lo, 0), min(hi, h_in)
    return (lo, hi) if hi > lo else (min(max(lo, 0), h_in), min(max(lo, 0), h_in))


class Band:
    """rows [g0, g0 + t.shape[1]) of a logical [1, H, W, C] map held by this rank; rows [v0, v1) of them are valid.
    k: log2 of the map's row scale relative to the coarsest (y latent) grid.  full: EVERY rank holds the whole map
    (valid everywhere) -- no exchange is needed to cut a slab from it."""

    def __init__(self, ctx, t, g0, H, k, v0, v1, full=False):
        self.ctx, self.t, self.g0, self.H, self.k, self.v0, self.v1, self.full = ctx, t, g0, H, k, v0, v1, full

    @property
    def shape(self):  # what the layers read: (n, rows, w, c) of the LOGICAL map
        return (1, self.H) + tuple(self.t.shape[2:])

    @property
    def device(self):
        return self.t.device

    def rows(self, a, b):
        """view of frame rows [a, b) (must be held)"""
        assert self.g0 <= a <= b <= self.g0 + self.t.shape[1], (self.g0, self.t.shape, a, b)
        return self.t[:, a - self.g0:b - self.g0]


class BandImages:
    """the first analysis layer's input: up to 3 images, each either a dict of WHOLE uint8 planes (every rank has the
    frames) or a float Band [.., 4] (the prediction) -- the banded twin of ops.ImageStack"""

    def __init__(self, ctx, parts, h, w, k):
        self.ctx, self.parts, self.H, self.w, self.k = ctx, list(parts), h, w, k
        self.shape = (1, h, w, 4 * len(self.parts))
        self.device = ctx.device
        self._aivc_cmap = tuple(4 * i + c for i in range(len(self.parts)) for c in range(3))


# ---- communication ------------------------------------------------------------------------------------------------
class DistComm:
    """the R ranks of a torch.distributed group (RCCL: device tensors, point-to-point over xGMI; gloo: staged through
    host memory -- the CPU-side test path)"""

    def __init__(self, group, ranks, local):
        import torch.distributed as dist
        self.dist, self.pg, self.ranks, self.r, self.R = dist, group, list(ranks), local, len(ranks)
        self.host = dist.get_backend(group) != 'nccl'
        self.stats = {'exchanges': 0, 'bytes_sent': 0, 'gathers': 0, 'bytes_gathered': 0}

    def exchange(self, sends, recvs):
        """sends: [(dst local rank, tensor)], recvs: [(src local rank, tensor view to fill)] -- the same plan on every
        rank (each derives everyone's needs from the partition), so the point-to-point operations pair up"""
        d = self.dist
        if not sends and not recvs:
            return
        self.stats['exchanges'] += 1
        self.stats['bytes_sent'] += sum(t.numel() * t.element_size() for _, t in sends)
        p2p, staged = [], []
        for dst, t in sends:
            buf = t.contiguous().cpu() if self.host else t.contiguous()
            p2p.append(d.P2POp(d.isend, buf, self.ranks[dst], group=self.pg))
        for src, view in recvs:
            buf = torch.empty(view.shape, dtype=view.dtype) if self.host else (view if view.is_contiguous() else torch.empty_like(view))
            staged.append((view, buf))
            p2p.append(d.P2POp(d.irecv, buf, self.ranks[src], group=self.pg))
        from . import parallel
        parallel._note('halo exchange (%d sends, %d receives)' % (len(sends), len(recvs)))
        with parallel._host_wait('halo exchange'):
            for w in d.batch_isend_irecv(p2p):
                w.wait()
        for view, buf in staged:
            if buf is not view:
                view.copy_(buf)

    def all_gather(self, t):
        """t: same shape on every rank -> [R, *t.shape]"""
        d = self.dist
        self.stats['gathers'] += 1
        self.stats['bytes_gathered'] += t.numel() * t.element_size() * (self.R - 1)
        src = t.contiguous().cpu() if self.host else t.contiguous()
        out = torch.empty((self.R,) + tuple(t.shape), dtype=t.dtype, device=src.device)
        from . import parallel
        parallel._note('all_gather band rows %s' % (tuple(t.shape),))
        with parallel._host_wait('all_gather of band rows'):
            d.all_gather_into_tensor(out.view(-1), src.view(-1), group=self.pg)
        return out.to(t.device)


class ThreadComm:
    """R virtual ranks = R threads of ONE process on one GPU (tests; also shows the scheme's launch / byte counts on a
    single-GPU box): the same banded code runs in every thread, the 'exchange' hands tensors over in memory.  All
    threads launch on the device's default stream, so the hand-over is stream ordered."""

    class Shared:
        def __init__(self, R):
            self.R, self.barrier, self.box = R, threading.Barrier(R), {}

    def __init__(self, shared, r):
        self.sh, self.r, self.R = shared, r, shared.R
        self.stats = {'exchanges': 0, 'bytes_sent': 0, 'gathers': 0, 'bytes_gathered': 0}

    def exchange(self, sends, recvs):
        self.stats['exchanges'] += bool(sends or recvs)
        self.stats['bytes_sent'] += sum(t.numel() * t.element_size() for _, t in sends)
        for dst, t in sends:
            self.sh.box[(self.r, dst)] = t
        self.sh.barrier.wait()
        for src, view in recvs:
            view.copy_(self.sh.box[(src, self.r)])
        self.sh.barrier.wait()
        for dst, _ in sends:
            self.sh.box.pop((self.r, dst), None)
        self.sh.barrier.wait()

    def all_gather(self, t):
        self.stats['gathers'] += 1
        self.stats['bytes_gathered'] += t.numel() * t.element_size() * (self.R - 1)
        self.sh.box[('g', self.r)] = t
        self.sh.barrier.wait()
        out = torch.stack([self.sh.box[('g', q)] for q in range(self.R)])
        self.sh.barrier.wait()
        self.sh.box.pop(('g', self.r), None)
        self.sh.barrier.wait()
        return out


# ---- the band engine ----------------------------------------------------------------------------------------------
class BandCtx:
    def __init__(self, comm, device):
        self.comm, self.device, self.r, self.R = comm, device, comm.r, comm.R
        self.yb = None
        self.launches = 0

    def set_frame(self, h_y, k_full):
        """rows of the coarsest grid (the y latent) of the frame being coded and the level of the frame itself
        (log2 of the analysis transform's reduction): fixes the partition at every level"""
        self.yb = [(i * h_y) // self.R for i in range(self.R + 1)]
        self.k_full = k_full

    def bounds(self, k, H):
        """partition of a map of H rows at level k (row scale 2^k relative to the y grid): R + 1 boundaries"""
        return [min(b << k, H) for b in self.yb[:-1]] + [H]

    def own(self, k, H):
        b = self.bounds(k, H)
        return b[self.r], b[self.r + 1]

    def full(self, t, k):
        """a map every rank holds entirely (all-gathered latents)"""
        return Band(self, t, 0, t.shape[1], k, 0, t.shape[1], full=True)

    # -- slabs ---------------------------------------------------------------------------------------------------
    def _plan(self, x, needs, in_bounds):
        """(sends, recv row ranges) for cutting slabs with per-rank needs [(lo, hi)] from a partitioned map"""
        sends, recvs = [], []
        if x.full:
            return sends, recvs
        lo_r, hi_r = needs[self.r]
        for q in range(self.R):
            if q == self.r:
                continue
            # rows of MY partition that q reads
            a, b = max(in_bounds[self.r], needs[q][0]), min(in_bounds[self.r + 1], needs[q][1])
            if b > a:
                sends.append((q, x.rows(a, b)))
            # rows of q's partition that I read
            a, b = max(in_bounds[q], lo_r), min(in_bounds[q + 1], hi_r)
            if b > a:
                recvs.append((q, (a, b)))
        return sends, recvs

    def slab(self, x, a, b, needs):
        """tensor of rows [a, b) of x: own valid rows copied, the others' fetched (needs: every rank's (lo, hi) --
        the rows that MUST be right; a .. lo may be unfilled: only discarded outputs read them)"""
        sends, recv_rows = self._plan(x, needs, self.bounds(x.k, x.H))
        out = None
        if b > a:
            lo, hi = needs[self.r]
            held = x.g0 <= a and b <= x.g0 + x.t.shape[1]
            if x.full or (not recv_rows and held and x.v0 <= lo and hi <= x.v1):
                out = x.rows(a, b)  # everything this rank reads is held and valid: a view, no copy
            else:
                out = torch.empty((1, b - a) + tuple(x.t.shape[2:]), dtype=x.t.dtype, device=x.t.device)
                c0, c1 = max(a, x.v0), min(b, x.v1)
                if c1 > c0:
                    out[:, c0 - a:c1 - a].copy_(x.rows(c0, c1))
        self.comm.exchange(sends, [(q, out[:, r0 - a:r1 - a]) for q, (r0, r1) in recv_rows])
        return out

    def align(self, band, g0, rows):
        """epilogue operand (residual / gate) as a tensor whose row 0 is frame row g0, `rows` rows long: a view when the
        band covers the range, else a copy of the overlap (rows outside feed discarded outputs only)"""
        if band is None:
            return None
        if not isinstance(band, Band):
            raise TypeError('band mode: epilogue operands must be bands')
        t = band.t
        if band.g0 <= g0 and g0 + rows <= band.g0 + t.shape[1]:
            return band.rows(g0, g0 + rows)
        out = torch.zeros((1, rows) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
        c0, c1 = max(g0, band.g0), min(g0 + rows, band.g0 + t.shape[1])
        if c1 > c0:
            out[:, c0 - g0:c1 - g0].copy_(band.rows(c0, c1))
        return out

    # -- one conv layer ------------------------------------------------------------------------------------------
    def conv(self, launch, x, mode, k, stride, pad, c_out, res=None, mul=None):
        """launch(x_slab, res_slab, mul_slab) -> y tensor runs the unchanged kernels on this rank's slab.
        x: Band or BandImages; c_out: channels of the result.  -> Band of the output (valid on this rank's rows of
        the output partition)."""
        h_in = x.H
        h_out = conv_out_rows(mode, h_in, k, stride, pad)
        k_out = x.k - 1 if (mode == abi.MODE_CONV and stride == 2) else (x.k + 1 if mode == abi.MODE_TCONV else x.k)
        ob = self.bounds(k_out, h_out)
        needs = [need_rows(mode, k, stride, pad, ob[q], ob[q + 1], h_in) for q in range(self.R)]
        o0, o1 = ob[self.r], ob[self.r + 1]
        lo, hi = needs[self.r]
        # first slab row: on a multiple of the stride (conv), so that local output j is frame output j + a / stride
        a = (lo // stride) * stride if mode == abi.MODE_CONV else lo
        if isinstance(x, BandImages):
            a = (lo // 2) * 2  # 4:2:0 planes: chroma row = luma row / 2
            parts = []
            for p in x.parts:
                if isinstance(p, Band):
                    parts.append(self.slab(p, a, hi, needs))
                elif p is None or hi <= a:
                    parts.append(None)
                else:
                    hc = (hi + 1) // 2
                    parts.append({'y': p['y'][:, a:hi], 'u': p['u'][:, a // 2:hc], 'v': p['v'][:, a // 2:hc]})
            xs = ops.ImageStack(parts, hi - a, x.w, self.device) if hi > a else None
        else:
            xs = self.slab(x, a, hi, needs)
        if o1 <= o0:  # nothing of this map is this rank's (it has served the others' halos above)
            w_in = x.w if isinstance(x, BandImages) else x.t.shape[2]
            w_out = conv_out_rows(mode, w_in, k, stride, pad)
            return Band(self, torch.empty((1, 0, w_out, c_out), device=self.device), o0, h_out, k_out, o0, o0)
        g0 = a // stride if mode == abi.MODE_CONV else (2 * a if mode == abi.MODE_TCONV else a)
        rows_out = conv_out_rows(mode, hi - a, k, stride, pad)
        y = launch(xs, self.align(res, g0, rows_out), self.align(mul, g0, rows_out))
        self.launches += 1
        assert y.shape[1] == rows_out and g0 <= o0 and o1 <= g0 + rows_out, (y.shape, rows_out, g0, o0, o1)
        return Band(self, y, g0, h_out, k_out, o0, o1)

    # -- whole maps <-> bands ------------------------------------------------------------------------------------
    def gather_full(self, band):
        """every rank's own rows of a banded map -> the whole map [1, H, W, C] on every rank"""
        b = self.bounds(band.k, band.H)
        m = max(b[q + 1] - b[q] for q in range(self.R))
        w, c = band.t.shape[2], band.t.shape[3]
        send = torch.zeros((m, w, c), dtype=torch.float32, device=self.device)
        n = b[self.r + 1] - b[self.r]
        if n:
            send[:n].copy_(band.rows(b[self.r], b[self.r + 1])[0])
        allr = self.comm.all_gather(send)
        return torch.cat([allr[q, :b[q + 1] - b[q]] for q in range(self.R)], dim=0).unsqueeze(0)

    def gather_planes(self, planes, h, w):
        """this rank's rows of the reconstructed 8-bit planes ({'y','u','v'} [1, rows, .]) -> whole planes everywhere"""
        b = self.bounds(self.k_full, h)
        hc, wc = (h + 1) // 2, (w + 1) // 2
        cb = [(x + 1) // 2 for x in b]
        m, mc = max(b[q + 1] - b[q] for q in range(self.R)), max(cb[q + 1] - cb[q] for q in range(self.R))
        send = torch.zeros(m * w + 2 * mc * wc, dtype=torch.uint8, device=self.device)
        n, nc = b[self.r + 1] - b[self.r], cb[self.r + 1] - cb[self.r]
        if n:
            send[:n * w].copy_(planes['y'].reshape(-1))
            send[m * w:m * w + nc * wc].copy_(planes['u'].reshape(-1))
            send[m * w + mc * wc:m * w + mc * wc + nc * wc].copy_(planes['v'].reshape(-1))
        allr = self.comm.all_gather(send)
        out = {}
        out['y'] = torch.cat([allr[q, :(b[q + 1] - b[q]) * w] for q in range(self.R)]).view(1, h, w)
        out['u'] = torch.cat([allr[q, m * w:m * w + (cb[q + 1] - cb[q]) * wc] for q in range(self.R)]).view(1, hc, wc)
        out['v'] = torch.cat([allr[q, m * w + mc * wc:m * w + mc * wc + (cb[q + 1] - cb[q]) * wc] for q in range(self.R)]).view(1, hc, wc)
        return out


def count_down(transform):
    """number of stride-2 stages of an analysis transform (its output grid is 2^-n of its input's)"""
    n = 0
    for m in transform:
        mode = getattr(m, 'mode', None)
        if mode == 'down':
            n += 1
        elif hasattr(m, 'layers') and len(m.layers) > 1 and getattr(m.layers[1], 'stride', (1,))[0] == 2:
            n += 1
    return n
