#!/bin/bash
# Round 3, third GPU call: where does the WaveFlow layer kernel's time go?  counters (two --pmc passes), memory ablations
# (PK_WF_ABLATE: wrong results, timing only), and HBM traffic (FETCH_SIZE / WRITE_SIZE passes).
set -u
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 200 python tools/quick_wf.py 64 > $OUT/quick_wf_64.log 2>&1; head -1 $OUT/quick_wf_64.log; grep wf_layer $OUT/quick_wf_64.log
for abl in 1 4 8 9 13; do
  PK_WF_ABLATE=$abl timeout 200 python tools/quick_wf_noassert.py 64 > $OUT/quick_wf_64_abl$abl.log 2>&1
  echo "ABL $abl: $(grep wf_layer $OUT/quick_wf_64_abl$abl.log)"
done
for act in 4 6; do
  PK_WF_ACTIVE=$act timeout 200 python tools/quick_wf.py 64 > $OUT/quick_wf_64_act$act.log 2>&1; echo "ACTIVE $act: $(grep wf_layer $OUT/quick_wf_64_act$act.log)"
done
cd /tmp
pmc() { timeout 240 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py wf 8 > $OUT/pmc_$1.log 2>&1; }
pmc A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc B "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"
pmc C "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum"
pmc D "WRITE_SIZE TCC_REQ_sum"
python $R/tools/pmc_parse.py $OUT/pmc_A $OUT/pmc_B $OUT/pmc_C $OUT/pmc_D --kernel=k_wf_ > $OUT/pmc_wf.json
find $OUT -type d -name "pmc_*" | xargs rm -rf
python - <<PY
import json
d = json.load(open("$OUT/pmc_wf.json"))
for k, v in d.items():
    if "layer" in k:
        print(k)
        for kk, vv in sorted(v.items()): print("   ", kk, round(vv, 1))
PY
ls $OUT
