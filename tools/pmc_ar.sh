#!/bin/bash
# PMC passes for the autoregressive models' kernels (run on the GPU box).  usage: tools/pmc_ar.sh <tag> [tts|taco] [steps]
set -u
TAG=${1:-arpmc}
M=${2:-taco}
STEPS=${3:-100}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() { timeout 150 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/quick_ar.py $M 32 $STEPS > $OUT/pmc_$1.log 2>&1; }
pmc A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
pmc B "FETCH_SIZE TCC_HIT"
pmc C "WRITE_SIZE TCC_MISS TCC_REQ"
python $R/tools/pmc_parse.py $OUT/pmc_A $OUT/pmc_B $OUT/pmc_C > $OUT/pmc_$M.json
python - <<PY
import json
d = json.load(open("$OUT/pmc_$M.json"))
for k, v in d.items():
    if v.get("_launches", 0) < 50: continue
    print(k, {kk: round(vv, 1) for kk, vv in sorted(v.items())})
PY
rm -rf $OUT/pmc_A $OUT/pmc_B $OUT/pmc_C
