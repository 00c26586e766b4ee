#!/usr/bin/env python3
"""Timeline of the last <window_s> seconds of a rocprofv3 kernel trace in <bin_ms> bins: busy fraction of the main queue
(the busiest one) and the number of range-encoder / range-decoder / other side-queue kernels running.
usage: trace_timeline.py <kernel_trace.csv> <window_s> [bin_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2])
B = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 25e6
qn = collections.Counter(r['Queue_Id'] for r in rows)
qmain = qn.most_common(1)[0][0]
end = max(int(r['End_Timestamp']) for r in rows)
w0 = end - int(win * 1e9)
bins = collections.defaultdict(lambda: [0, 0, 0, 0])
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e < w0:
        continue
    n = r['Kernel_Name']
    cls = 0 if r['Queue_Id'] == qmain else (1 if 'range_encode' in n else 2 if 'range_decode' in n else 3)
    for b in range(int((max(s, w0) - w0) // B), int((e - w0) // B) + 1):
        lo, hi = max(s, w0 + b * B), min(e, w0 + (b + 1) * B)
        if hi > lo:
            bins[b][cls] += hi - lo
for b in sorted(bins):
    v = bins[b]
    print('%6d ms  main %4.0f%%  enc %5.2f  dec %5.2f  other-side %5.2f' % (b * B / 1e6, 100 * v[0] / B, v[1] / B, v[2] / B, v[3] / B))
