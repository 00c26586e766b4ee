#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
t0=$(date +%s)
timeout 2700 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/t_full.log 2>&1
echo "full gpu suite rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -25 gpurun_out/t_full.log
python -c "import __graft_entry__ as g; g.smoke()"
