# tuning aid (GPU box): phase decomposition of conv_images (tools/build_images_exp.sh variants), with and without GDN
for v in base nomfma nostage noepi onlymfma onlystage; do
  for n in 1 2; do
    echo "$v: $(AIVC_HIP_LIB=$PWD/aivc_amd/lib/exp/img_$v.so BATCH=32 python tools/conv_images_probe.py $n 5 2>/dev/null | tail -1)"
  done
done
