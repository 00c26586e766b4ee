#!/bin/bash
# Tuning aid: build alternative conv_mfma variants into aivc_amd/lib/exp/<name>.so (select with AIVC_HIP_LIB).
# usage: [SRC=conv_wino] tools/build_exp.sh name "-DAIVC_BK=64 -DAIVC_EXP_PIPE" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build_hip()" >/dev/null
mkdir -p aivc_amd/lib/exp
SRC=${SRC:-conv_mfma}
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Iinclude -Iaivc_amd/csrc"
while [ $# -ge 2 ]; do
  ( /opt/rocm/bin/hipcc $F $2 -c aivc_amd/csrc/$SRC.hip -o aivc_amd/lib/exp/$1.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o aivc_amd/lib/exp/$1.so aivc_amd/lib/exp/$1.o \
      $(ls aivc_amd/lib/obj/*.o | grep -v "/$SRC.o") && echo built $1 ) &
  shift 2
done
wait
