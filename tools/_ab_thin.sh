for m in 1 4 8 1 8; do
  AIVC_THIN_GRID_MULT=$m AIVC_LAYER_TABLE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$m.json 2> gpurun_out/ab_$m.txt
  echo "mult $m: $(python -c "import json;d=json.load(open('gpurun_out/ab_$m.json'));print(d['ms_per_step'], d['value'], d['roofline']['per_variant']['thin_mfma_kernel'])")"
done
