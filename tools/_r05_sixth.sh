#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "range or coder or entropy" > gpurun_out/t_rc.log 2>&1
echo "rc tests rc=$?"; tail -5 gpurun_out/t_rc.log
timeout 200 python tools/bench_rangecoder.py 2>&1 | grep "streams  1\|streams 64" | tee gpurun_out/rc_dec2.txt
timeout 600 python -m pytest tests/test_gpu_codec.py tests/test_decoder_golden.py -x -q -m gpu > gpurun_out/t_codec.log 2>&1
echo "codec rc=$?"; tail -3 gpurun_out/t_codec.log
timeout 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
python tools/show_bench.py gpurun_out/bench_a.json 2>/dev/null | head -1
python -c "import json; d=json.load(open('gpurun_out/bench_a.json')); print('HR', d['high_rate'])"
