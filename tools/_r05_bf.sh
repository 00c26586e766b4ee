#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_precision.py -x -q -m gpu -s > gpurun_out/t_prec.log 2>&1
echo "precision rc=$?"; grep -v "^$" gpurun_out/t_prec.log | tail -45
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_family" > gpurun_out/t_conv.log 2>&1
echo "conv family rc=$?"; tail -2 gpurun_out/t_conv.log
timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-high-rate > gpurun_out/bench_bf.json 2> gpurun_out/bench_bf.err
tail -3 gpurun_out/bench_bf.err
python tools/show_bench.py gpurun_out/bench_bf.json 2>/dev/null | head -1
python -c "
import json; d=json.load(open('gpurun_out/bench_bf.json'))['precision_mode']
print({k:d[k] for k in ('value','ms_per_step','vs_headline','closed_loop_ok','stream_errors')})
print(d['bf16x3_kernels']['fp32_equivalent_tflops'], d['bf16x3_kernels']['ms_per_step'], d['fp32_contract_kernels_left'])
for k,v in d['bf16x3_kernels']['per_variant'].items(): print('  ',k,v)
"
