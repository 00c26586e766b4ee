#!/bin/bash
# GPU box: first analysis layer (aivc_conv_images) under alternative builds in aivc_amd/lib/exp/.  usage: LIBS="a b" tools/ab_images.sh
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
for rep in 1 2 3; do
for lib in ${LIBS:-old new}; do
  if [ $lib != new ]; then export AIVC_HIP_LIB=$root/aivc_amd/lib/exp/$lib.so; else unset AIVC_HIP_LIB; fi
  for n in 1 2; do
    echo -n "$lib: "; BATCH=32 timeout 120 python tools/conv_images_probe.py $n 8 2>&1 | tail -1
  done
done
done
