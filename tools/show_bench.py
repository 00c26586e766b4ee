#!/usr/bin/env python3
"""Pretty-print the JSON line of bench.py (file argument, else stdin)."""
import json
import sys
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    print('value %.3f %s | enc %.2f dec %.2f fps | ms/step %.1f | closed_loop %s | bytes/frame %.0f' % (
        d['value'], d['unit'], d.get('encode_main_stream_fps_rank0', d.get('encode_fps_rank0', 0)), d.get('decode_main_stream_fps_rank0', d.get('decode_fps_rank0', 0)), d['ms_per_step'],
        d.get('closed_loop_ok'), d.get('bytes_per_frame', 0)))
    r = d.get('roofline')
    if r:
        print('roofline', r['kernel'], r['achieved'], r['unit'], 'frac', r['frac'], '| all mfma conv', r['all_mfma_conv'])
        for k, v in r['per_variant'].items():
            print('   %-34s %s' % (k, v))
    if d.get('cpu_baseline'):
        print('cpu_baseline', d['cpu_baseline'])
    for key in ('high_rate', 'bitstream_only_encoder', 'precision_mode'):
        o = d.get(key)
        if o:
            print(key, {k: o[k] for k in ('value', 'ms_per_step', 'vs_headline', 'encode_main_stream_fps', 'decode_main_stream_fps',
                                           'bytes_equal_full_encoder', 'closed_loop_ok', 'closed_loop_ok_on_references') if k in o})
