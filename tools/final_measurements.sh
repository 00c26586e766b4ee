root=${GRAFT_REPO_ROOT:-$PWD}
cd $root; mkdir -p gpurun_out; R=r06
timeout 1200 python bench.py > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench_line.err
echo "bench rc=$?"
python tools/show_bench.py gpurun_out/${R}_bench_line.json 2>/dev/null | head -30
python -c "
import json; d=json.load(open('gpurun_out/${R}_bench_line.json')); print('contract_v2', {k: d['contract_v2'][k] for k in ('value','ms_per_step','vs_headline','closed_loop_ok')}, d['contract_v2']['winograd_kernel']); print('pipelined', d['pipelined']['value'], 'hr pipelined', d['high_rate']['pipelined']['value'])"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${R}_kstats -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${R}_bench_line_under_rocprof.json 2> $root/gpurun_out/${R}_under_rocprof.err
cd $root
f=$(find gpurun_out/${R}_kstats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${R}_kernel_stats_bench_1080p.csv; head -8 gpurun_out/${R}_kernel_stats_bench_1080p.csv | cut -c1-150
rm -rf gpurun_out/${R}_kstats
python -c "
import json; d=json.load(open('gpurun_out/${R}_bench_line_under_rocprof.json')); print('under rocprof', d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline']['launches'])"
