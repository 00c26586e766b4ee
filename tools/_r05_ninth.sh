#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "range or coder or entropy" > gpurun_out/t_rc.log 2>&1
echo "rc tests rc=$?"; tail -3 gpurun_out/t_rc.log
timeout 200 python tools/bench_rangecoder.py 2>&1 | grep "streams  1\|streams 64 maps 64" | tee gpurun_out/rc_dec4.txt
timeout 600 python -m pytest tests/test_gpu_codec.py tests/test_decoder_golden.py -x -q -m gpu > gpurun_out/t_codec.log 2>&1
echo "codec rc=$?"; tail -3 gpurun_out/t_codec.log
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
python tools/show_bench.py gpurun_out/bench_a.json 2>/dev/null | head -1
python -c "import json; d=json.load(open('gpurun_out/bench_a.json')); print('HR', {k:d['high_rate'][k] for k in ('value','ms_per_step','encode_main_stream_fps','decode_main_stream_fps','vs_headline','closed_loop_ok')})"
timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-high-rate --width 3840 --height 2160 --frames 32 --active-y 64,64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K HR', {k: d[k] for k in ('value','ms_per_step','encode_main_stream_fps_rank0','decode_main_stream_fps_rank0','closed_loop_ok')})"
