#!/bin/bash
# round 5, second GPU call: scheduling change validated, full default bench (new cpu baseline + high_rate), 4K points
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_codec.py tests/test_decoder_golden.py tests/test_gpu_multi_process.py -x -q --durations=5 -m gpu > gpurun_out/t_codec.log 2>&1
echo "codec rc=$?"
tail -12 gpurun_out/t_codec.log
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_full.json'))
print({k:d[k] for k in ('value','ms_per_step','encode_main_stream_fps_rank0','decode_main_stream_fps_rank0','closed_loop_ok','stream_errors_rank0','parity_checked')})
print('HR', d['high_rate'])
c=d['cpu_baseline']; print('CPU', {k:c[k] for k in ('value','cores','frames','encode_fps','decode_fps','closed_loop')}, c['one_core'], c['fmaf_oracle'])
PY
run() {
  line=$(timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-high-rate "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','coded_frames_per_s','ms_per_step','encode_main_stream_fps_rank0','decode_main_stream_fps_rank0','closed_loop_ok','bytes_per_frame')})")
  echo "$* $line"
}
run --width 3840 --height 2160 --frames 32
run --width 3840 --height 2160 --frames 32 --active-y 64,64
run --width 3840 --height 2160 --frames 32 --active-y 64,64 --entropy-lookahead 2
run --active-y 64,64 --entropy-lookahead 2
