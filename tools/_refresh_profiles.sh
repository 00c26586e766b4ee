set -x
root=$PWD
timeout 600 python bench.py 2>gpurun_out/bench_stderr.txt > gpurun_out/r03_bench_line.json; python tools/show_bench.py gpurun_out/r03_bench_line.json | head -3
timeout 900 tools/other_configs.sh gpurun_out/r03_other_configs.txt > /dev/null 2>&1; cat gpurun_out/r03_other_configs.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/r03_bench_line_under_rocprof.json 2>/dev/null
s=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
if [ -n "$s" ]; then cp $s $root/gpurun_out/r03_kernel_stats_bench_1080p.csv; fi
if [ -n "$t" ]; then cd $root && timeout 300 python tools/trace_gaps.py $t 3.3 > gpurun_out/r03_main_queue_gaps.txt 2>&1; head -12 gpurun_out/r03_main_queue_gaps.txt; fi
