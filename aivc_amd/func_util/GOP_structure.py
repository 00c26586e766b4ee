"""GOP dependency graphs (src/func_util/GOP_structure.py): {frame_i: {type, prev_ref, next_ref,
coding_order}} with i the display index.  Random-access GOPs are hierarchical-B, numbered
depth-first exactly like the reference; chained GOPs share one I frame; LDP is an I + P chain."""
FRAME_I = 0
FRAME_P = 1
FRAME_B = 2


def _name(i):
    return 'frame_' + str(i)


def generate_ra_gop(gop_size):
    gop = {_name(0): {'type': FRAME_I, 'prev_ref': None, 'next_ref': None, 'coding_order': 0},
           _name(gop_size): {'type': FRAME_P, 'prev_ref': _name(0), 'next_ref': None, 'coding_order': 1}}
    order = [2]

    def split(mid, half):
        gop[_name(mid)] = {'type': FRAME_B, 'prev_ref': _name(mid - half), 'next_ref': _name(mid + half),
                           'coding_order': order[0]}
        order[0] += 1
        half //= 2
        if half:
            split(mid - half, half)
            split(mid + half, half)
    split(gop_size // 2, gop_size // 2)
    return gop


def chained_gop(gop_size, n):
    """n hierarchical GOPs behind a single I frame; GOP i is shifted by i * gop_size in display
    index, reference names and coding order."""
    out = generate_ra_gop(gop_size)
    for i in range(1, n):
        off = i * gop_size
        for fname, d in generate_ra_gop(gop_size).items():
            idx = int(fname.split('_')[-1])
            if idx == 0:
                continue

            def sh(r):
                return None if r is None else _name(int(r.split('_')[-1]) + off)
            out[_name(idx + off)] = {'type': d['type'], 'prev_ref': sh(d['prev_ref']),
                                     'next_ref': sh(d['next_ref']), 'coding_order': d['coding_order'] + off}
    return out


def generate_ldp_gop(gop_size):
    gop = {_name(0): {'type': FRAME_I, 'prev_ref': None, 'next_ref': None, 'coding_order': 0}}
    for i in range(1, gop_size + 1):
        gop[_name(i)] = {'type': FRAME_P, 'prev_ref': _name(i - 1), 'next_ref': None, 'coding_order': i}
    return gop


def get_name_frame_code(GOP_struct, idx_code):
    return [f for f in GOP_struct if GOP_struct[f].get('coding_order') == idx_code]


def get_depth_gop(GOP_struct):
    return max(d.get('coding_order') for d in GOP_struct.values())


def generate_gop_struct(gop_struct_name: str):
    """'1_GOP_0' (all intra), 'LDP_<n>', '<k>_GOP_<n>'."""
    toks = gop_struct_name.split('_')
    if gop_struct_name == '1_GOP_0':
        return {_name(0): {'type': FRAME_I, 'prev_ref': None, 'next_ref': None, 'coding_order': 0}}
    if 'LDP' in toks:
        return generate_ldp_gop(int(toks[-1]))
    return chained_gop(int(toks[-1]), int(toks[0]))


def coding_levels(GOP_struct):
    """Frames grouped by dependency depth (breadth-first schedule): frames of one level only depend
    on earlier levels, so they can be coded concurrently.  The bitstream stores frames in display
    order, so any topological order produces the same bytes."""
    depth = {}

    def d(f):
        if f not in depth:
            refs = [r for r in (GOP_struct[f]['prev_ref'], GOP_struct[f]['next_ref']) if r is not None]
            depth[f] = 0 if not refs else 1 + max(d(r) for r in refs)
        return depth[f]
    for f in GOP_struct:
        d(f)
    levels = {}
    for f, v in depth.items():
        levels.setdefault(v, []).append(f)
    return [sorted(levels[k], key=lambda s: int(s.split('_')[-1])) for k in sorted(levels)]
