#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout 200 python tools/_dbg_thin.py 2>&1 | grep "==" 
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_wide_golden.py tests/test_gpu_fullsize_kernels.py tests/test_gpu_reference_layers.py -x -q -m gpu > gpurun_out/t_ops.log 2>&1
echo "ops rc=$?"; tail -4 gpurun_out/t_ops.log
BATCHES=16,32 timeout 150 python tools/thin_probe.py 2>&1 | tail -4
timeout 700 bash tools/pmc_conv.sh r05b_thin 15 BATCH=16 > gpurun_out/pmc_thin_b.log 2>&1
python - <<'PY'
import json
try:
    c=list(json.load(open('gpurun_out/pmc_r05b_thin.json')).values())[0]
    mf=c['SQ_INSTS_MFMA']
    print('thin pmc: busy %.3f valu/mfma %.2f salu/mfma %.2f lds/mfma %.2f lds conflict %.0f of %.0f wait_inst_any/wave_cycles %.2f'%(32*mf/(1024*c['GRBM_GUI_ACTIVE']/8),(c['SQ_INSTS_VALU']-mf)/mf,c['SQ_INSTS_SALU']/mf,c['SQ_INSTS_LDS']/mf,c['SQ_LDS_BANK_CONFLICT'],c['SQ_LDS_IDX_ACTIVE'],c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']))
    print(c)
except Exception as e: print('pmc failed', e)
PY
