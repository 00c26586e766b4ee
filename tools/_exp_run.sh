#!/bin/bash
# runs on the GPU box: bench_conv over the prebuilt variants in aivc_amd/lib/exp/
for v in "$@"; do
  echo "=== $v"
  AIVC_HIP_LIB=$PWD/aivc_amd/lib/exp/$v.so BATCH=8 timeout 300 python tools/bench_conv.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids
done
