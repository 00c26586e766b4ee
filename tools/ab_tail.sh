#!/bin/bash
# GPU box: the fused 1x1-tail kernel under alternative builds in aivc_amd/lib/exp/.  usage: LIBS="a b" tools/ab_tail.sh
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
for rep in 1 2 3; do
for lib in ${LIBS:-old new}; do
  if [ $lib != new ]; then export AIVC_HIP_LIB=$root/aivc_amd/lib/exp/$lib.so; else unset AIVC_HIP_LIB; fi
  for nb in 64 16 4; do echo -n "$lib: "; BATCH=$nb timeout 120 python tools/tail_probe.py 10 2>&1 | tail -1; done
done
done
