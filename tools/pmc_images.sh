#!/bin/bash
# GPU box: SQ counters of aivc_conv_images (tools/conv_images_probe.py), separate rocprofv3 --pmc passes (never combined
# with tracing).  usage: tools/pmc_images.sh <tag> <n_img>  -> gpurun_out/pmc_<tag>.json
tag=$1; nimg=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python $root/tools/conv_images_probe.py $nimg 3 > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_images' in r.get('Kernel_Name', ''):
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in per.items():
        res[c] = sum(v) / len(v)
if 'SQ_VALU_MFMA_BUSY_CYCLES' in res and 'GRBM_GUI_ACTIVE' in res:
    res['mfma_busy_frac'] = res['SQ_VALU_MFMA_BUSY_CYCLES'] / (res['GRBM_GUI_ACTIVE'] * 1024.0 / 8.0)
if res.get('SQ_INSTS_MFMA'):
    for k in ('VALU', 'LDS', 'SALU', 'VMEM_RD'):
        if 'SQ_INSTS_' + k in res:
            res[k.lower() + '_per_mfma'] = (res['SQ_INSTS_' + k] - (res['SQ_INSTS_MFMA'] if k == 'VALU' else 0)) / res['SQ_INSTS_MFMA']
json.dump(res, open(out + '.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
