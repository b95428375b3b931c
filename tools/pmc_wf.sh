#!/bin/bash
# PMC passes for the WaveFlow layer kernel (run on the GPU box).  usage: tools/pmc_wf.sh <tag>
set -u
TAG=${1:-wfpmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() { timeout 240 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py wf 8 > $OUT/pmc_$1.log 2>&1; }
pmc A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc B "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"
python $R/tools/pmc_parse.py $OUT/pmc_A $OUT/pmc_B --kernel=k_wf_ > $OUT/pmc_wf.json
python - <<PY
import json
d = json.load(open("$OUT/pmc_wf.json"))
for k, v in d.items():
    print(k)
    for kk, vv in sorted(v.items()): print("   ", kk, vv)
PY
