#!/bin/bash
# full -m gpu suite, smoke, the default bench line, the same under rocprofv3 --kernel-trace --stats, layer A/B tools: one box
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; mkdir -p gpurun_out; R=r06
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/${R}_suite.txt; cat gpurun_out/${R}_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python bench.py > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench_line.err; echo "bench rc=$?"
python tools/show_bench.py gpurun_out/${R}_bench_line.json 2>/dev/null | head -24
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${R}_kstats -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${R}_bench_line_under_rocprof.json 2> $root/gpurun_out/${R}_under_rocprof.err
cd $root
f=$(find gpurun_out/${R}_kstats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${R}_kernel_stats_bench_1080p.csv; head -6 gpurun_out/${R}_kernel_stats_bench_1080p.csv | cut -c1-150
rm -rf gpurun_out/${R}_kstats
BATCH=16 python tools/bench_poly.py > gpurun_out/${R}_ab_polyphase.txt 2>/dev/null; BATCH=64 python tools/bench_poly.py >> gpurun_out/${R}_ab_polyphase.txt 2>/dev/null; cat gpurun_out/${R}_ab_polyphase.txt
