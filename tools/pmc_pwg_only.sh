#!/bin/bash
# The three PMC passes of the Parallel WaveGAN layer kernel alone -> pwg_layer_traffic.json (see tools/round_last_call.sh)
set -u
TAG=${1:-r03pmcp}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() { timeout 100 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py $2 > $OUT/pmc_$1.log 2>&1; }
pmc pA "pwg 32" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc pB "pwg 32" "FETCH_SIZE TCC_HIT"
pmc pC "pwg 32" "WRITE_SIZE TCC_MISS TCC_REQ"
python $R/tools/pmc_parse.py $OUT/pmc_pA $OUT/pmc_pB $OUT/pmc_pC --kernel=k_pwg_ > $OUT/pmc_pwg.json
python $R/tools/pmc_traffic.py pwg $OUT/pmc_pwg.json $OUT/pwg_layer_traffic.json
find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
