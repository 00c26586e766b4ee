#!/bin/bash
# GPU box: instruction counters of the range coder kernels (one rocprofv3 --pmc pass, no tracing) on 64 streams of
# 522 240 symbols at 2.3 bit per symbol.  usage: tools/pmc_rangecoder.sh <tag>  -> gpurun_out/pmc_<tag>.json
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ONLY=64,64,1.5 timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python $root/tools/bench_rangecoder.py > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.OrderedDict()
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'range_' not in k:
            continue
        per[k.split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in per.items():
        for c, v in d.items():
            res.setdefault(k, {})[c] = sum(v) / len(v)
sym = 64 * 522240.0
for k, d in res.items():
    d['per_symbol'] = {c: round(v / sym, 2) for c, v in d.items() if c.startswith('SQ_INSTS')}
json.dump(res, open(out + '.json', 'w'), indent=1)
print(json.dumps({k: v.get('per_symbol') for k, v in res.items()}, indent=1))
PY
rm -rf $out
