# tuning aid (GPU box): first analysis layer with / without its fused GDN, inside the bench
for v in "" 1 "" 1; do
  AIVC_IMAGES_UNFUSED_GDN=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/abi_$v.json 2>/dev/null
  echo "unfused=[$v]: $(python -c "import json;d=json.load(open('gpurun_out/abi_$v.json'));r=d['roofline']['per_variant'];print(d['ms_per_step'], d['value'], {k:v['ms_total'] for k,v in r.items() if 'images' in k or 'gdn,' in k or k.startswith('conv_mfma<gdn')})")"
done
