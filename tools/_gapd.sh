root=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
if [ -n "$t" ]; then cd $root && timeout 300 python tools/trace_gap_detail.py $t 3.3 0.5 > gpurun_out/gap_detail.txt 2>&1; fi
