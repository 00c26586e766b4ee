"""In-memory frame / GOP / video codec on device-resident tensors: the orchestration the
reference spreads over FullNet.GOP_forward (missing from the snapshot), Decoder.decode
(src/real_life/decode.py:455-580), decode_one_GOP (:193-327) and infer_one_sequence
(src/model_mngt/model_management.py:31-244) -- without PNG round trips, temp files or per-latent
device->host CDF copies.  Frames are dicts {'y','u','v'} of uint8 CUDA planes [1,h,w] (8-bit
references are exact: the reference casts every reconstruction to 8-bit levels, decode.py:575).
"""
import math

import torch

from . import ops
from .func_util.GOP_structure import FRAME_B, FRAME_I, FRAME_P, generate_gop_struct
from .real_life import cat_binary_files as container
from .real_life import header as hdr
from .real_life.bitstream import finalize_frame, split_sections


def frame_index(name):
    return int(name.split('_')[-1])


class FrameCodec:
    def __init__(self, full_net):
        self.net = full_net
        self.mof = full_net.mode_net.mode_net
        self.cod = full_net.codec_net.codec_net

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def to444(planes, h, w, device):
        if planes is None:
            return torch.zeros((1, h, w, 3), dtype=torch.float32, device=device)
        return ops.yuv420_to_444(planes['y'], planes['u'], planes['v'], c_store=3)

    def _motion(self, mof_out, prev444, next444, h, w, frame_type, want_aux):
        return ops.warp_blend(mof_out, prev444, next444, h, w, frame_type, co=3, want_aux=want_aux)

    def encode_frame(self, cur, prev, nxt, frame_type, idx_rate=0., want_aux=False):
        """-> {'bytes', 'rec' (uint8 planes), 'data_dim', + aux tensors when want_aux}"""
        dev = cur['y'].device
        h, w = cur['y'].shape[-2:]
        code = self.to444(cur, h, w, dev)
        sections = [None] * 4
        pred = skip = None
        aux = {}
        if frame_type != FRAME_I:
            prev444 = self.to444(prev, h, w, dev)
            next444 = self.to444(nxt if frame_type == FRAME_B else None, h, w, dev)
            a = self.mof.analyse(torch.cat((code, prev444, next444), dim=3), frame_type, idx_rate)
            short_in = torch.cat((prev444, next444), dim=3) if frame_type == FRAME_B else None
            mof_out = self.mof.synthesise(a['y_hat'], short_in)
            wb = self._motion(mof_out, prev444, next444, h, w, frame_type, want_aux)
            pred, skip = wb['pred'], wb['skip']
            sections[0] = self.mof.ac.pend_z(a['q_z'])
            sections[1] = self.mof.ac.pend_y(a['q_y'], a['sigma'])
            if want_aux:
                aux.update(alpha=wb['alpha'], beta=wb['beta'], warping=wb['x_warp'])
        zero_pred = torch.zeros_like(code) if pred is None else pred
        c = self.cod.analyse(torch.cat((code, zero_pred), dim=3), frame_type, idx_rate)
        cod_out = self.cod.synthesise(c['y_hat'], pred)
        _, rec8 = ops.frame_to_yuv420(cod_out, h, w, skip=skip, want_float=False)
        sections[2] = self.cod.ac.pend_z(c['q_z'])
        sections[3] = self.cod.ac.pend_y(c['q_y'], c['sigma'])
        data_dim = {'x': (h, w), 'y': c['dim_y'], 'z': c['dim_z'],
                    'x_uv': (math.ceil(h / 2), math.ceil(w / 2))}
        out = {'bytes': finalize_frame(sections), 'rec': dict(zip('yuv', rec8)), 'data_dim': data_dim}
        if want_aux:
            aux['code'] = code
            out['aux'] = aux
        return out

    def _decode_net(self, net, payload_z, payload_y, frame_type, data_dim, in_shortcut, idx_rate, device):
        h_y, w_y = data_dim['y']
        h_z, w_z = data_dim['z']
        q_z = net.ac.decode_z(payload_z, 1, h_z, w_z, net.nb_ft_z, device)
        y_hat = net.latents_from_symbols(q_z, lambda sigma: net.ac.decode_y(payload_y, sigma), frame_type,
                                         (h_y, w_y), idx_rate)
        return net.synthesise(y_hat, in_shortcut)

    def decode_frame(self, frame_bytes, prev, nxt, frame_type, data_dim, idx_rate=0., device=None):
        """Mirror of Decoder.decode (src/real_life/decode.py:455-580) -> uint8 planes dict."""
        device = device or torch.device('cuda')
        h, w = data_dim['x']
        sec = split_sections(frame_bytes)
        pred = skip = None
        if frame_type != FRAME_I:
            prev444 = self.to444(prev, h, w, device)
            next444 = self.to444(nxt if frame_type == FRAME_B else None, h, w, device)
            short_in = torch.cat((prev444, next444), dim=3) if frame_type == FRAME_B else None
            mof_out = self._decode_net(self.mof, sec[0], sec[1], frame_type, data_dim, short_in, idx_rate, device)
            wb = self._motion(mof_out, prev444, next444, h, w, frame_type, False)
            pred, skip = wb['pred'], wb['skip']
        cod_out = self._decode_net(self.cod, sec[2], sec[3], frame_type, data_dim, pred, idx_rate, device)
        _, rec8 = ops.frame_to_yuv420(cod_out, h, w, skip=skip, want_float=False)
        return dict(zip('yuv', rec8))

    # ------------------------------------------------------------------------------------------
    def encode_gop(self, frames, gop_name, idx_rate=0.):
        """frames: list (display order) of uint8 plane dicts, len == len(GOP struct).
        -> (gop bytes, reconstructions in display order, data_dim)"""
        gop = generate_gop_struct(gop_name)
        order = sorted(gop, key=lambda f: gop[f]['coding_order'])
        rec, fbytes, data_dim = {}, {}, None
        for f in order:
            d = gop[f]
            out = self.encode_frame(frames[frame_index(f)], rec.get(d['prev_ref']), rec.get(d['next_ref']),
                                    d['type'], idx_rate)
            rec[f], fbytes[f], data_dim = out['rec'], out['bytes'], out['data_dim']
        names = sorted(gop, key=frame_index)
        blob = container.pack_gop(hdr.gop_header_bytes(gop_name, idx_rate), [fbytes[f] for f in names])
        return blob, [rec[f] for f in names], data_dim

    def decode_gop(self, gop_bytes, data_dim, device=None):
        gop_name, idx_rate, fbytes = container.unpack_gop(gop_bytes)
        gop = generate_gop_struct(gop_name)
        order = sorted(gop, key=lambda f: gop[f]['coding_order'])
        rec = {}
        for f in order:
            d = gop[f]
            rec[f] = self.decode_frame(fbytes[frame_index(f)], rec.get(d['prev_ref']), rec.get(d['next_ref']),
                                       d['type'], data_dim, idx_rate, device)
        return [rec[f] for f in sorted(gop, key=frame_index)]

    # ------------------------------------------------------------------------------------------
    def encode_video(self, frames, gop_name, idx_starting_frame=0, idx_end_frame=None, idx_rate=0.,
                     unit_filter=None):
        """frames[i] is the frame with absolute index idx_starting_frame + i.  The last intra-period
        unit is padded by repeating the last frame (src/model_mngt/model_management.py:142-153).
        unit_filter(u) -> bool selects the units this process codes (multi-GPU sharding); skipped
        units come back as None in the returned list of GOP blobs."""
        n = len(frames)
        idx_end_frame = idx_starting_frame + n - 1 if idx_end_frame is None else idx_end_frame
        unit = len(generate_gop_struct(gop_name))
        nb_gop = math.ceil(n / unit)
        gops, recs, data_dim = [], [], None
        for u in range(nb_gop):
            if unit_filter is not None and not unit_filter(u):
                gops.append(None)
                recs.append(None)
                continue
            chunk = [frames[min(u * unit + i, n - 1)] for i in range(unit)]
            blob, rec, data_dim = self.encode_gop(chunk, gop_name, idx_rate)
            gops.append(blob)
            recs.append(rec)
        return {'gops': gops, 'recs': recs, 'data_dim': data_dim, 'nb_gop': nb_gop,
                'idx_starting_frame': idx_starting_frame, 'idx_end_frame': idx_end_frame}

    @staticmethod
    def assemble_video(enc):
        vh = hdr.video_header_bytes(enc['data_dim'], enc['nb_gop'], enc['idx_starting_frame'],
                                    enc['idx_end_frame'])
        return container.pack_video(vh, enc['gops'])

    def decode_video(self, blob, device=None, unit_filter=None):
        """-> list of uint8 plane dicts for frames idx_first..idx_last (padded frames removed)."""
        data_dim, first, last, gops = container.unpack_video(blob)
        frames = []
        for u, g in enumerate(gops):
            if unit_filter is not None and not unit_filter(u):
                name, _, fb = container.unpack_gop(g)
                frames.extend([None] * len(fb))
                continue
            frames.extend(self.decode_gop(g, data_dim, device))
        return frames[:last - first + 1], data_dim, first, last
