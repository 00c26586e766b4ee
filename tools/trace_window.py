#!/usr/bin/env python3
"""Kernels of the last <window_s> seconds of a rocprofv3 kernel trace, per queue class (busiest queue = the codec's main
stream, the rest = entropy side streams): calls, total and average time.  usage: trace_window.py <kernel_trace.csv> <window_s>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2])
qn = collections.Counter(r['Queue_Id'] for r in rows)
qmain = qn.most_common(1)[0][0]
end = max(int(r['End_Timestamp']) for r in rows)
w0 = end - int(win * 1e9)
for cls in ('main', 'side'):
    acc = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        if (r['Queue_Id'] == qmain) != (cls == 'main') or int(r['Start_Timestamp']) < w0:
            continue
        a = acc[r['Kernel_Name'].split('(')[0][-70:]]
        a[0] += 1
        a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(v[1] for v in acc.values())
    print('== %s queue(s): %d kernels, %.1f ms in the last %.2f s' % (cls, sum(v[0] for v in acc.values()), tot / 1e6, win))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
        print('  %-70s %6d %9.2f ms %8.1f us' % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
