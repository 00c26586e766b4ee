#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>.json (tools/pmc_conv.sh: raw means per counter) -> the summary kept under profiles/: per-MFMA instruction
ratios, matrix-pipe busy fraction, L2 hit rate, memory-side bytes per launch with the guide's gfx950 correction (FETCH_SIZE
doubled, KB -> bytes).  usage: pmc_summary.py <raw.json> <out.json> <kernel label> <probe text> <algorithmic bytes per launch>"""
import json
import sys


def main():
    raw, out, label, probe, alg = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5])
    d = json.load(open(raw))
    name, c = max(d.items(), key=lambda kv: kv[1].get('SQ_INSTS_MFMA', 0))
    mf = c['SQ_INSTS_MFMA']
    traffic = int((2.0 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024)
    res = {
        'kernel': label, 'kernel_symbol': name, 'probe': probe,
        'collection': 'tools/pmc_conv.sh: seven separate rocprofv3 --pmc passes (SQ x4, FETCH_SIZE, WRITE_SIZE, TCC hit / miss), never combined with tracing; mean over the 3 launches of each pass',
        'sq_counters': {
            'mfma_busy_frac': round(c['mfma_busy_frac'], 4),
            'mfma_busy_note': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)',
            'SQ_INSTS_MFMA': mf,
            'non_mfma_valu_per_mfma': round((c['SQ_INSTS_VALU'] - mf) / mf, 3),
            'salu_per_mfma': round(c['SQ_INSTS_SALU'] / mf, 3),
            'lds_per_mfma': round(c['SQ_INSTS_LDS'] / mf, 3),
            'vmem_rd_per_mfma': round(c['SQ_INSTS_VMEM_RD'] / mf, 3),
            'SQ_LDS_BANK_CONFLICT': c['SQ_LDS_BANK_CONFLICT'], 'SQ_LDS_IDX_ACTIVE': c['SQ_LDS_IDX_ACTIVE'],
            'wave_quad_cycles': {k: c[k] for k in ('SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS')},
        },
        'mfma_gflop_executed': round(mf * 4096 / 1e9, 1) if 'thin' not in name else round(mf * 2048 / 1e9, 1),
        'l2_hit_rate': round(c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']), 3),
        'fetch_size_kb': c['FETCH_SIZE'], 'write_size_kb': c['WRITE_SIZE'],
        'correction': 'FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section; Infinity-Cache hits are counted, so this is an upper bound on HBM proper); KB -> bytes x1024',
        'traffic_bytes_per_launch': traffic, 'algorithmic_bytes_per_launch': int(alg),
        'traffic_over_algorithmic': round(traffic / alg, 2),
    }
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res['sq_counters'], indent=None)[:400], res['traffic_over_algorithmic'], res['l2_hit_rate'])


if __name__ == '__main__':
    main()
