for v in "$@"; do
  echo "=== $v"
  AIVC_HIP_LIB=$PWD/aivc_amd/lib/exp/$v.so BATCH=8 timeout 200 python tools/bench_conv.py 2>&1 | grep -v "^$"
done
