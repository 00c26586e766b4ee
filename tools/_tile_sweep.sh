#!/bin/bash
# GPU box: tools/bench_conv.py with every forced tile (AIVC_FORCE_TILE), one "=== tileN" section each
for t in auto 0 1 2 4 5; do
  echo "=== tile$t"
  if [ $t = auto ]; then BATCH=8 timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
  else AIVC_FORCE_TILE=$t BATCH=8 timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids; fi
done
