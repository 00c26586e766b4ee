#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "range or coder or entropy" > gpurun_out/t_rc.log 2>&1
echo "rc tests rc=$?"; tail -3 gpurun_out/t_rc.log
timeout 200 python tools/bench_rangecoder.py 2>&1 | grep "streams  1" | tee gpurun_out/rc_dec3.txt
timeout 600 python -m pytest tests/test_gpu_codec.py tests/test_decoder_golden.py -x -q -m gpu > gpurun_out/t_codec.log 2>&1
echo "codec rc=$?"; tail -3 gpurun_out/t_codec.log
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
python tools/show_bench.py gpurun_out/bench_a.json 2>/dev/null | head -1
python -c "import json; d=json.load(open('gpurun_out/bench_a.json')); print('HR', {k:d['high_rate'][k] for k in ('value','ms_per_step','encode_main_stream_fps','decode_main_stream_fps','vs_headline','closed_loop_ok')})"
AIVC_RC_DEC_LOWPRIO=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_lp.json 2> gpurun_out/bench_lp.err
python tools/show_bench.py gpurun_out/bench_lp.json 2>/dev/null | head -1
python -c "import json; d=json.load(open('gpurun_out/bench_lp.json')); print('HR lowprio', {k:d['high_rate'][k] for k in ('value','ms_per_step','encode_main_stream_fps','decode_main_stream_fps','vs_headline','closed_loop_ok')})"
timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-high-rate --width 3840 --height 2160 --frames 32 --active-y 64,64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K HR', {k: d[k] for k in ('value','ms_per_step','encode_main_stream_fps_rank0','decode_main_stream_fps_rank0','closed_loop_ok')})"
cd /tmp && export TMPDIR=/tmp
AIVC_NO_QUALITY=1 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof4k -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-high-rate --width 3840 --height 2160 --frames 32 --active-y 64,64 > $root/gpurun_out/prof4k.log 2>&1
cd $root
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/prof4k/**/*kernel_stats.csv', recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:14]: print('%-90s %6s %9.1f ms avg %8.1f us'%(r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
