#!/bin/bash
# GPU box: SQ / MFMA-busy counters of ONE conv shape (tools/conv_probe.py), separate rocprofv3 --pmc passes
# (never combined with tracing).  usage: tools/pmc_conv.sh <tag> <shape-index> [env assignments...]
# -> gpurun_out/pmc_<tag>.json
tag=$1; idx=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python $root/tools/conv_probe.py $idx 3 > $out/p$i.log 2>&1
done
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = collections.OrderedDict()
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r.get('Kernel_Name', '')
        if 'conv_mfma' not in k and 'thin' not in k and 'conv_wino' not in k and 'gdn_resident' not in k:
            continue
        per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in per.items():
        short = k.split('(')[0][:90]
        for c, v in d.items():
            res.setdefault(short, {})[c] = sum(v) / len(v)  # mean over dispatches
for k, d in res.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d:
        d['mfma_busy_frac'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] * 1024.0 / 8.0)  # 1024 SIMDs; GRBM_GUI_ACTIVE arrives summed over the 8 XCDs
json.dump(res, open(out + '.json', 'w'), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
