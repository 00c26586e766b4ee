"""ArithmeticCoder: CDF build + torchac-compatible range coding + per-frame section framing
(src/real_life/bitstream.py), entirely on the GPU.

Frame bitstream = 4 sections in fixed order mofnet_z, mofnet_y, codecnet_z, codecnet_y; each is
[n_bytes 4 B BE][payload]; y payloads start with [n_nonzero_maps 1 B][map index 1 B each]; an I
frame carries two empty (4 zero bytes) MOFNet sections; an all-zero y is the single byte 0.

What differs from the reference implementation (not from its bytes): the [C,H,W,514] fp32 CDF is
never materialised -- the encoder evaluates the 2 CDF points a symbol needs, the decoder reads a
uint16 row per position that another kernel produced, z uses a [C_z,514] table built once; there is
no device->host copy of CDFs and no temp file.
"""
import os

import numpy as np
import torch

from .. import abi, ops
from ..func_util.nn_util import get_value
from .utils import BITSTREAM_SUFFIX

SECTION_NAMES = ('mofnet_z', 'mofnet_y', 'codecnet_z', 'codecnet_y')


class PendingSection:
    """A latent whose symbols are known on the device but not yet range-coded."""

    def __init__(self, mode, q, sigma=None, table=None):
        self.mode, self.q, self.sigma, self.table = mode, q, sigma, table
        self.flags = ops.nonzero_flags(q) if mode == 'laplace' else None  # async, no sync here


def split_sections(frame_bytes):
    """frame bytes -> 4 payloads (bytes)."""
    out, pos = [], 0
    for _ in SECTION_NAMES:
        n = int.from_bytes(frame_bytes[pos:pos + 4], 'big')
        out.append(frame_bytes[pos + 4:pos + 4 + n])
        pos += 4 + n
    return out


def finalize_frame(sections):
    """sections: list of 4 PendingSection or None (None = empty section).  Runs the CDF-bound and
    range-encode kernels (one wavefront per section, concurrently) and returns the frame bytes."""
    heads, jobs = [None] * 4, []
    for i, s in enumerate(sections):
        if s is None:
            continue
        if s.mode == 'laplace':
            flags = s.flags.cpu().numpy()  # the only host sync of the entropy stage: C bytes
            maps = [int(c) for c in np.nonzero(flags)[0]]
            heads[i] = bytes([len(maps)]) + bytes(maps)
            if maps:
                jobs.append((i, ops.laplace_bounds(s.sigma, s.q, maps)))
        else:
            heads[i] = b''
            jobs.append((i, ops.table_bounds(s.table, s.q)))
    payload = [b''] * 4
    if jobs:
        out, lens, offs = ops.range_encode([b for _, b in jobs])
        out_h, lens_h = out.cpu().numpy(), lens.cpu().numpy()
        for (i, _), (off, _cap), ln in zip(jobs, offs, lens_h):
            if int(ln) < 0:
                raise RuntimeError('range encoder output buffer overflow')
            payload[i] = out_h[off:off + int(ln)].tobytes()
    frame = b''
    for i in range(4):
        if sections[i] is None:
            frame += (0).to_bytes(4, 'big')
        else:
            body = heads[i] + payload[i]
            frame += len(body).to_bytes(4, 'big') + body
    return frame


class ArithmeticCoder():
    def __init__(self, param):
        default = {'balle_pdf_estim_z': None, 'device': 'cpu', 'AC_MAX_VAL': abi.AC_MAX_VAL}
        self.balle_pdf_estim = get_value('balle_pdf_estim_z', param, default)
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

    # ---- in-memory API (device tensors in NHWC) -----------------------------------------------
    def pend_z(self, q_z):
        return PendingSection('pmf', q_z, table=self.z_table(q_z.device))

    def pend_y(self, q_y, sigma):
        return PendingSection('laplace', q_y, sigma=sigma)

    def decode_z(self, payload, n, h, w, c, device):
        """payload bytes -> q_z int16 NHWC [n,h,w,c] (pmf mode, all channels)."""
        sym = ops.range_decode([payload], [self.z_table(device)], [c * n * h * w], [n * h * w])[0]
        return ops.scatter_symbols(sym, n * h * w, c, list(range(c))).view(n, h, w, c)

    def decode_y(self, payload, sigma):
        """payload bytes + sigma NHWC -> q_y int16 NHWC (zero maps restored)."""
        n, h, w, c = sigma.shape
        n_maps = payload[0]
        maps = list(payload[1:1 + n_maps])
        npix = n * h * w
        if n_maps == 0:
            return torch.zeros((n, h, w, c), dtype=torch.int16, device=sigma.device)
        rows = ops.laplace_cdf_rows(sigma, maps)
        sym = ops.range_decode([payload[1 + n_maps:]], [rows], [n_maps * npix], [0])[0]
        return ops.scatter_symbols(sym, npix, c, maps).view(n, h, w, c)

    # ---- path-based API with the reference's signatures (NCHW float tensors, one file per frame) --
    def encode(self, param):
        default = {'x': None, 'mode': 'laplace', 'sigma': None, 'bitstream_path': None, 'flag_debug': True,
                   'latent_name': '', 'flag_md5sum': False}
        x = get_value('x', param, default)
        mode = get_value('mode', param, default)
        sigma = get_value('sigma', param, default)
        path = get_value('bitstream_path', param, default)
        latent_name = get_value('latent_name', param, default)
        if get_value('flag_md5sum', param, default):
            raise NotImplementedError('flag_md5sum debug sections are not implemented')
        if not path.endswith(BITSTREAM_SUFFIX):
            path += BITSTREAM_SUFFIX
        q = ops.to_nhwc(x).to(torch.int16)
        sec = self.pend_y(q, ops.to_nhwc(sigma)) if mode == 'laplace' else self.pend_z(q)
        slots = [None] * 4
        slots[SECTION_NAMES.index(latent_name)] = sec
        body = split_sections(finalize_frame(slots))[SECTION_NAMES.index(latent_name)]
        blob = len(body).to_bytes(4, 'big') + body
        if latent_name == 'codecnet_z' and not os.path.isfile(path):
            blob = (0).to_bytes(8, 'big') + blob
        with open(path, 'ab') as f:
            f.write(blob)

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
            payload = split_sections(f.read())[SECTION_NAMES.index(latent_name)]
        if mode == 'laplace':
            q = self.decode_y(payload, ops.to_nhwc(sigma))
        else:
            b, c, h, w = data_dim
            q = self.decode_z(payload, b, h, w, c, torch.device(device))
        return ops.to_nchw_view(q.float())
