#!/bin/bash
# GPU box: kernel trace of two timed headline steps (no extras), then where the main queue idles and what ran.
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-high-rate --no-precision-mode --no-lean-encoder --no-roofline > /tmp/kt_line.json 2> /tmp/kt_err.txt
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
cd $root
python tools/show_bench.py /tmp/kt_line.json 2>/dev/null | head -5; cut -c1-300 /tmp/kt_line.json
ms=$(python -c "import json;print(json.load(open('/tmp/kt_line.json'))['ms_per_step']/1e3*2)")
echo "window $ms s"
python tools/trace_gaps.py $f $ms
python tools/trace_gap_detail.py $f $ms 1.0 | head -80
python tools/trace_window.py $f $ms | head -75
