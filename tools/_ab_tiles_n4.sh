# tuning aid (GPU box): tile menu at the batch of the narrow levels (4 frames), low-resolution layers
for t in auto 0 1 5 6; do
  if [ $t = auto ]; then unset AIVC_FORCE_TILE; else export AIVC_FORCE_TILE=$t; fi
  echo "== tile $t"
  BATCH=${BATCH:-4} python tools/bench_conv.py 2>/dev/null | grep -E "cheng conv3 128 @135p|att 3x3|res 3x3 128 @68p|gs1 tconv5|gs2 tconv3|att 1x1|res 1x1|ga4" | grep -v "algo=2"
done
