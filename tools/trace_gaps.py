#!/usr/bin/env python3
"""Idle gaps of the busiest HIP queue in a rocprofv3 kernel trace.  usage: trace_gaps.py <kernel_trace.csv> [window_s]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
qn = collections.Counter(r['Queue_Id'] for r in rows)
qmain = qn.most_common(1)[0][0]
print('queues', dict(qn), 'main', qmain)
main = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if r['Queue_Id'] == qmain)
w1 = main[-1][1] - int(0.05e9)
w0 = w1 - int(win * 1e9)
sel = [m for m in main if m[0] >= w0 and m[1] <= w1]
busy = sum(e - s for s, e, _ in sel)
print('window %.1f s: %d kernels, busy %.3f s' % (win, len(sel), busy / 1e9))
gaps, gapn, hist, tot = collections.Counter(), collections.Counter(), collections.Counter(), 0
for (s0, e0, n0), (s1, e1, n1) in zip(sel, sel[1:]):
    g = s1 - e0
    if g > 0:
        tot += g
        hist['<5us' if g < 5e3 else '<20us' if g < 20e3 else '<100us' if g < 100e3 else '<1ms' if g < 1e6 else '>=1ms'] += g
        if g > 20e3:
            gaps[(n0[:48], n1[:48])] += g
            gapn[(n0[:48], n1[:48])] += 1
print('total gap %.3f s' % (tot / 1e9), {k: round(v / 1e6, 1) for k, v in hist.items()})
for k, v in gaps.most_common(14):
    print('%8.1f ms n=%4d  %s -> %s' % (v / 1e6, gapn[k], k[0], k[1]))
