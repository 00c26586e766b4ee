"""Same kernel, same shape: mean launch time inside the encoder against inside the decoder of one 1080p clip (the decoder's
entropy stage keeps range-coder waves resident on some SIMDs: a persistent kernel with a static share per CU waits for its
slowest CU).  HIP events around every conv launch (ops.PROFILE)."""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from aivc_amd import ops, synth  # noqa: E402
from aivc_amd.models import arch  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    active = [int(v) for v in os.environ.get('ACTIVE_Y', '6,12').split(',')]
    model = synth.make_model(arch.DEFAULT_WIDTHS, seed=1234, device=dev)
    synth.calibrate_operating_point(model, dev, active_mof=active[0], active_cod=active[1]) if 'active_mof' in synth.calibrate_operating_point.__code__.co_varnames else synth.calibrate_operating_point(model, dev)
    fc = model.frame_codec()
    fc.max_batch = 64
    units = [bench.gpu_synthetic_unit(1920, 1080, 33, 33 * u, dev, 7 + u) for u in range(4)]
    with torch.no_grad():
        blobs, _, dd = fc.encode_units(units, '1_GOP_32')
        fc.decode_units(blobs, dd, dev)
        torch.cuda.synchronize()
        out = {}
        for phase in ('enc', 'dec'):
            ops.PROFILE = []
            if phase == 'enc':
                blobs, _, dd = fc.encode_units(units, '1_GOP_32')
            else:
                fc.decode_units(blobs, dd, dev)
            torch.cuda.synchronize()
            acc = defaultdict(lambda: [0, 0.0])
            for variant, flops, e0, e1, shape in ops.PROFILE:
                a = acc[(variant,) + tuple(shape)]
                a[0] += 1
                a[1] += e0.elapsed_time(e1)
            ops.PROFILE = None
            out[phase] = acc
    rows = []
    for k in out['enc']:
        if k in out['dec']:
            ne, te = out['enc'][k]
            nd, td = out['dec'][k]
            rows.append((td / nd / (te / ne), bench.variant_name(k[0]), k[1:], ne, te / ne, nd, td / nd, td))
    rows.sort(key=lambda r: -r[7])
    tot_e = sum(v[1] for v in out['enc'].values())
    tot_d = sum(v[1] for v in out['dec'].values())
    print('conv launches: encoder %.1f ms, decoder %.1f ms' % (tot_e, tot_d))
    extra = 0.0
    for ratio, name, shape, ne, me, nd, md, td in rows[:40]:
        extra += (md - me) * nd
        print('x%.2f  %-32s %-44s enc %3d x %7.3f ms | dec %3d x %7.3f ms' % (ratio, name, shape, ne, me, nd, md))
    print('decoder time above the encoder\'s rate for the same launches: %.1f ms' % sum((r[6] - r[4]) * r[5] for r in rows))


if __name__ == '__main__':
    main()
