#!/usr/bin/env python3
"""Per-kernel totals and main-queue gaps of the LAST `window_s` seconds of a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes NAME_results.db on this ROCm).
usage: trace_db.py <results.db> [window_s] [csv_out]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 3.4
rows = db.execute('select name, start, end, queue_id, stream_id from kernels order by start').fetchall()
t1 = max(r[2] for r in rows) - int(0.05e9)
t0 = t1 - int(win * 1e9)
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]


def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '').replace('aivc::', '')
    return n[:70]


tot = collections.defaultdict(lambda: [0, 0])
for n, s, e, q, st in sel:
    d = tot[short(n)]
    d[0] += 1
    d[1] += e - s
qn = collections.Counter(r[4] for r in sel)
main = qn.most_common(1)[0][0]
busy_all = sum(v[1] for v in tot.values())
print('window %.2f s: %d kernels on %d streams, sum of kernel time %.3f s' % (win, len(sel), len(qn), busy_all / 1e9))
lines = ['kernel,calls,total_ms,avg_us,percent_of_window']
for k, (c, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    lines.append('"%s",%d,%.3f,%.2f,%.2f' % (k, c, ns / 1e6, ns / c / 1e3, 100.0 * ns / (win * 1e9)))
print('\n'.join(lines[:45]))
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write('\n'.join(lines) + '\n')
m = sorted((s, e, n) for n, s, e, q, st in sel if st == main)
busy = sum(e - s for s, e, _ in m)
gap = collections.Counter()
gapn = collections.Counter()
tg = 0
for (s0, e0, n0), (s1, e1, n1) in zip(m, m[1:]):
    g = s1 - e0
    if g > 0:
        tg += g
        if g > 20e3:
            gap[(short(n0)[:40], short(n1)[:40])] += g
            gapn[(short(n0)[:40], short(n1)[:40])] += 1
print('main stream %s: %d kernels, busy %.3f s, gaps %.3f s' % (main, len(m), busy / 1e9, tg / 1e9))
for k, v in gap.most_common(10):
    print('%8.1f ms n=%4d  %s -> %s' % (v / 1e6, gapn[k], k[0], k[1]))
