#!/bin/bash
# counters of the kernels of the round's second session on one box + the other configurations + widths under the final tree
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root; mkdir -p gpurun_out; R=r06
timeout 900 bash tools/pmc_conv.sh ${R}_wino_poly5 2 BATCH=16 > gpurun_out/pmc_poly.log 2>&1; tail -3 gpurun_out/pmc_poly.log      # 5x5 s2 64->128 @540x960: conv_wino_kernel<1>
timeout 900 bash tools/pmc_conv.sh ${R}_wino_tconv5 14 BATCH=16 > gpurun_out/pmc_tc.log 2>&1; tail -3 gpurun_out/pmc_tc.log        # tconv 5x5 128->64 @270x480: conv_wino_kernel<2>
n=$(python -c "import sys; sys.path.insert(0,'tools'); import conv_probe; print(len(conv_probe.PROBES) - 1)")
timeout 900 bash tools/pmc_conv.sh ${R}_wino $n BATCH=16 > gpurun_out/pmc_wino.log 2>&1; tail -3 gpurun_out/pmc_wino.log
timeout 900 bash tools/pmc_conv.sh ${R}_wino_68x120 11 BATCH=64 > gpurun_out/pmc_wino68.log 2>&1; tail -3 gpurun_out/pmc_wino68.log
timeout 900 bash tools/pmc_conv.sh ${R}_gdn_resident 3 BATCH=32 > gpurun_out/pmc_gdn.log 2>&1; tail -3 gpurun_out/pmc_gdn.log
BATCH=16 python tools/bench_wino.py > gpurun_out/${R}_ab_winograd.txt 2>/dev/null; BATCH=64 python tools/bench_wino.py >> gpurun_out/${R}_ab_winograd.txt 2>/dev/null; cat gpurun_out/${R}_ab_winograd.txt
for wd in w192 w144; do timeout 600 python bench.py --widths $wd --steps 2 --warmup 1 --no-high-rate --no-lean-encoder --no-precision-mode --no-cpu-baseline --no-pipelined --no-contract-v2 > gpurun_out/${R}_widths_$wd.json 2>/dev/null; done
timeout 2400 bash tools/other_configs.sh gpurun_out/${R}_other_configs.txt
timeout 300 python tools/cli_wallclock.py gpurun_out/${R}_cli_wallclock.json 2>/dev/null | tail -1
