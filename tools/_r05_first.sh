#!/bin/bash
# round 5, first GPU call: new config tests, counters at HEAD for the two worst kernels, a baseline line
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q --durations=8 > gpurun_out/t_configs.log 2>&1
echo "configs rc=$?" 
tail -15 gpurun_out/t_configs.log
BATCH=16 bash tools/pmc_conv.sh r05_thin 15 BATCH=16 > gpurun_out/pmc_thin.log 2>&1
bash tools/pmc_images.sh r05_images1 1 > gpurun_out/pmc_images1.log 2>&1
bash tools/pmc_images.sh r05_images2 2 > gpurun_out/pmc_images2.log 2>&1
cd $root
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_head.json 2> gpurun_out/bench_head.err
python tools/show_bench.py gpurun_out/bench_head.json 2>/dev/null | head -40
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --active-y 64,64 > gpurun_out/bench_hr.json 2> gpurun_out/bench_hr.err
python -c "import json; d=json.load(open('gpurun_out/bench_hr.json')); print('HR', {k: d[k] for k in ('value','ms_per_step','encode_main_stream_fps_rank0','decode_main_stream_fps_rank0','bytes_per_frame','closed_loop_ok')})"
