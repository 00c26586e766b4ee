#!/usr/bin/env python3
"""Tabulate tools/_exp_run.sh logs (variant x shape TFLOP/s)."""
import re, sys, collections
cur = None; data = collections.OrderedDict(); order = []
for line in open(sys.argv[1]):
    if line.startswith('==='):
        cur = line.split()[1]; order.append(cur); continue
    m = re.match(r'(.{32}) algo=(\d) (\+gdn|    )\s+([\d.]+) ms\s+([\d.]+) GFLOP\s+([\d.]+) TFLOP', line)
    if m:
        data.setdefault((m.group(1).strip(), m.group(2), m.group(3).strip()), {})[cur] = float(m.group(6))
    elif line.startswith('sum'):
        data.setdefault(('SUM', '', ''), {})[cur] = float(line.split('->')[1].split()[0])
print('%-40s' % 'shape' + ''.join('%9s' % o for o in order))
for k, v in data.items():
    print('%-40s' % (' '.join(k)) + ''.join('%9.1f' % v.get(o, 0) for o in order))
