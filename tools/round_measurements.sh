#!/bin/bash
# the measurements DESIGN.md section 7 quotes, all on one box (files named r05_*: rename per round)
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python tools/show_bench.py gpurun_out/r05_bench_line.json 2>/dev/null | head -24
cd /tmp && export TMPDIR=/tmp
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r05_kstats -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/r05_bench_line_under_rocprof.json 2> $root/gpurun_out/r05_under_rocprof.err
cd $root
f=$(find gpurun_out/r05_kstats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05_kernel_stats_bench_1080p.csv; head -12 gpurun_out/r05_kernel_stats_bench_1080p.csv | cut -c1-150
timeout 900 bash tools/pmc_conv.sh r05_dom32 2 BATCH=32 FUSE_GDN=1 > gpurun_out/pmc_dom.log 2>&1
tail -3 gpurun_out/pmc_dom.log
timeout 1500 bash tools/other_configs.sh gpurun_out/r05_other_configs.txt
timeout 300 python tools/bench_rangecoder.py > gpurun_out/r05_rangecoder.txt 2>&1
tail -5 gpurun_out/r05_rangecoder.txt
