#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/lat_probe.hip -o /tmp/lat_probe && timeout 90 /tmp/lat_probe > gpurun_out/lat_probe.txt 2> gpurun_out/lat_probe.err
echo "lat_probe rc=$?"; cat gpurun_out/lat_probe.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_wide_golden.py tests/test_gpu_fullsize_kernels.py tests/test_gpu_reference_layers.py -x -q -m gpu > gpurun_out/t_ops.log 2>&1
echo "ops rc=$?"; tail -6 gpurun_out/t_ops.log
BATCHES=16,32 timeout 150 python tools/thin_probe.py 2>&1 | tail -6
echo "--- range coder, stream per lane"; timeout 200 python tools/bench_rangecoder.py 2>&1 | grep "streams  1\|streams 64" | tee gpurun_out/rc_lanes.txt
echo "--- range coder, wave per stream"; AIVC_RC_ENCODE=wave timeout 200 python tools/bench_rangecoder.py 2>&1 | grep "streams  1\|streams 64" | tee gpurun_out/rc_wave.txt
timeout 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
python tools/show_bench.py gpurun_out/bench_a.json 2>/dev/null | head -20
python -c "import json; d=json.load(open('gpurun_out/bench_a.json')); print('HR', d['high_rate'])"
