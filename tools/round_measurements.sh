#!/bin/bash
# the measurements DESIGN.md section 7 quotes, all on one box (round 6: files named r06_*)
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
R=r06
t0=$(date +%s)
timeout 1200 python bench.py > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench_line.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python tools/show_bench.py gpurun_out/${R}_bench_line.json 2>/dev/null | head -30
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${R}_kstats -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${R}_bench_line_under_rocprof.json 2> $root/gpurun_out/${R}_under_rocprof.err
cd $root
f=$(find gpurun_out/${R}_kstats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${R}_kernel_stats_bench_1080p.csv; head -14 gpurun_out/${R}_kernel_stats_bench_1080p.csv | cut -c1-150
rm -rf gpurun_out/${R}_kstats
timeout 900 bash tools/pmc_conv.sh ${R}_dom32 2 BATCH=32 FUSE_GDN=1 > gpurun_out/pmc_dom.log 2>&1
tail -3 gpurun_out/pmc_dom.log
n=$(python -c "import sys; sys.path.insert(0,'tools'); import conv_probe; print(len(conv_probe.PROBES) - 1)")
timeout 900 bash tools/pmc_conv.sh ${R}_wino $n BATCH=16 PRECISION=fp32w > gpurun_out/pmc_wino.log 2>&1
tail -3 gpurun_out/pmc_wino.log
timeout 900 bash tools/pmc_conv.sh ${R}_wino_68x120 11 BATCH=64 PRECISION=fp32w > gpurun_out/pmc_wino68.log 2>&1   # right-edge column in the masked fast epilogue
tail -3 gpurun_out/pmc_wino68.log
timeout 900 bash tools/pmc_conv.sh ${R}_gdn_resident 3 BATCH=32 > gpurun_out/pmc_gdn.log 2>&1                          # csrc/gdn.hip (variant 400)
tail -3 gpurun_out/pmc_gdn.log
python tools/bench_gdn.py 128 64 272 480 > gpurun_out/${R}_gdn.txt 2>/dev/null; python tools/bench_gdn.py 64 64 272 480 >> gpurun_out/${R}_gdn.txt 2>/dev/null; cat gpurun_out/${R}_gdn.txt
python tools/wino_sizes.py > gpurun_out/${R}_wino_sizes.txt 2>/dev/null; cat gpurun_out/${R}_wino_sizes.txt
python tools/enc_vs_dec_kernels.py 2>/dev/null | tail -3
BATCH=16 python tools/bench_wino.py > gpurun_out/${R}_ab_winograd.txt 2>/dev/null; BATCH=64 python tools/bench_wino.py >> gpurun_out/${R}_ab_winograd.txt 2>/dev/null; cat gpurun_out/${R}_ab_winograd.txt
timeout 300 python tools/cli_wallclock.py gpurun_out/${R}_cli_wallclock.json 2>/dev/null | tail -1
for wd in w192 w144; do timeout 600 python bench.py --widths $wd --steps 2 --warmup 1 --no-high-rate --no-lean-encoder --no-precision-mode --no-cpu-baseline --no-pipelined --no-contract-v2 > gpurun_out/${R}_widths_$wd.json 2>/dev/null; done
timeout 2400 bash tools/other_configs.sh gpurun_out/${R}_other_configs.txt
timeout 300 python tools/bench_rangecoder.py > gpurun_out/${R}_rangecoder.txt 2>&1
tail -5 gpurun_out/${R}_rangecoder.txt
