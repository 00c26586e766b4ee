#!/usr/bin/env python3
"""Every idle gap >= min_ms of the busiest HIP queue in a rocprofv3 kernel trace, with what the other queues were
running meanwhile.  usage: trace_gap_detail.py <kernel_trace.csv> [window_s] [min_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 3.3
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
qn = collections.Counter(r['Queue_Id'] for r in rows)
qmain = qn.most_common(1)[0][0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:], r['Queue_Id']) for r in rows)
main = [e for e in ev if e[3] == qmain]
other = [e for e in ev if e[3] != qmain]
w1 = main[-1][1] - int(0.05e9)
w0 = w1 - int(win * 1e9)
sel = [m for m in main if m[0] >= w0 and m[1] <= w1]
for (s0, e0, n0, _), (s1, e1, n1, _) in zip(sel, sel[1:]):
    g = s1 - e0
    if g >= min_ms * 1e6:
        print('t=%8.1f ms gap %6.2f ms  %s -> %s' % ((e0 - w0) / 1e6, g / 1e6, n0[-40:], n1[-40:]))
        act = [(max(s, e0), min(e, s1), n, q) for s, e, n, q in other if s < s1 and e > e0]
        for s, e, n, q in sorted(act)[:8]:
            print('      q%s %-44s overlaps %6.2f ms (kernel %.2f ms)' % (q, n[-44:], (e - s) / 1e6,
                  next((ee - ss) / 1e6 for ss, ee, nn, qq in other if nn == n and qq == q and ss <= s and ee >= e)))
