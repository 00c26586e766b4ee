#!/bin/bash
# Tuning aid: conv_images.hip with phase-decomposition switches -> aivc_amd/lib/exp/img_<name>.so (AIVC_HIP_LIB selects)
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build_hip()" >/dev/null
mkdir -p aivc_amd/lib/exp
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Iinclude -Iaivc_amd/csrc"
while [ $# -ge 2 ]; do
  ( /opt/rocm/bin/hipcc $F $2 -c aivc_amd/csrc/conv_images.hip -o aivc_amd/lib/exp/img_$1.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o aivc_amd/lib/exp/img_$1.so aivc_amd/lib/exp/img_$1.o \
      $(ls aivc_amd/lib/obj/*.o | grep -v conv_images) && echo built $1 ) &
  shift 2
done
wait
