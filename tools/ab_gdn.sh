#!/bin/bash
# GPU box: the fused-GDN epilogue A/B (lean sqrt / division against a baseline build in aivc_amd/lib/exp/old.so) + the
# tests that cover it.  usage: [LIBS="old new v6"] tools/ab_gdn.sh [skip-tests]
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
if [ -z "$1" ]; then
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_kernels.py tests/test_wide_golden.py tests/test_gpu_reference_layers.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1
  echo "kernel tests rc=$?"; tail -5 gpurun_out/ab_tests.log
  timeout 600 python -m pytest tests/test_gpu_codec.py -x -q -m gpu -k "bitstream_only or match_oracle or closed_loop" > gpurun_out/ab_codec.log 2>&1
  echo "codec tests rc=$?"; tail -3 gpurun_out/ab_codec.log
fi
for rep in 1 2; do
for lib in ${LIBS:-old new}; do
  if [ $lib != new ]; then export AIVC_HIP_LIB=$root/aivc_amd/lib/exp/$lib.so; else unset AIVC_HIP_LIB; fi
  [ -n "$AIVC_HIP_LIB" ] && [ ! -f "$AIVC_HIP_LIB" ] && continue
  echo "== $lib (rep $rep)"
  BATCH=16 timeout 300 python tools/bench_conv.py 2>&1 | grep -E "gdn|sum"
  BATCH=32 timeout 120 python tools/conv_images_probe.py 1 5 2>&1 | tail -1
  BATCH=32 timeout 120 python tools/conv_images_probe.py 2 5 2>&1 | tail -1
done
done
