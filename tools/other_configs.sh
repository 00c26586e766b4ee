#!/bin/bash
# GPU box: the other BASELINE.json configurations on the same box and binary as the headline line
# usage: tools/other_configs.sh <out.txt>
out=$1
echo "# python bench.py --no-cpu-baseline --no-roofline --no-high-rate --no-precision-mode --no-lean-encoder --no-contract-v2 <flags>, one MI355X, same box" > $out
run() {
  line=$(timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-high-rate --no-precision-mode --no-lean-encoder --no-contract-v2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r = {k: d[k] for k in ('value','coded_frames_per_s','ms_per_step','encode_main_stream_fps_rank0','decode_main_stream_fps_rank0','closed_loop_ok','stream_errors_rank0','bytes_per_frame')}; r['two_clips_in_flight'] = (d.get('pipelined') or {}).get('value'); r['two_clips_closed_loop_ok'] = (d.get('pipelined') or {}).get('closed_loop_ok'); print(r)")
  echo "$* $line" >> $out
}
run                                                       # configs[3] on one GPU (the headline workload)
run --active-y 64,64                                      # ... at HIGH rate: all 64 y maps of both networks non-zero
run --width 3840 --height 2160 --frames 32                # configs[4] at the low-rate calibration
run --width 3840 --height 2160 --frames 32 --active-y 64,64   # configs[4] at HIGH rate ("ms_ssim-2")
run --width 416 --height 240 --gop 1_GOP_0 --frames 64    # configs[1]
run --width 1280 --height 720 --gop LDP_8 --frames 64     # configs[2]
run --width 416 --height 240 --gop 2_GOP_16 --frames 101  # configs[0]'s shape (sanity_script.sh: RA gop 16, intra period 32, frames 0-100)
run --frames 32                                           # one 1080p unit: what one rank codes when configs[3] is spread over 4 GPUs
cat $out
