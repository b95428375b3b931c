#!/bin/bash
# Counter traffic of the PWG layer kernel collected FIRST (so the bench line's roofline.traffic is of this kernel source), then the bench as the driver
# runs it and its rocprofv3 kernel stats.   usage: tools/r06_traffic_bench_call.sh <tag>
set -u
TAG=${1:-r06w}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PK_ROUND="round 6"
cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc() { timeout 240 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py $2 > $OUT/pmc_$1.log 2>&1; }
pmc pA "pwg 32" "$SQ"; pmc pB "pwg 32" "FETCH_SIZE TCC_HIT"; pmc pC "pwg 32" "WRITE_SIZE TCC_MISS TCC_REQ"
python $R/tools/pmc_parse.py $OUT/pmc_pA $OUT/pmc_pB $OUT/pmc_pC --kernel=k_pwg_ > $OUT/pmc_pwg.json
python $R/tools/pmc_traffic.py pwg $OUT/pmc_pwg.json $OUT/pwg_layer_traffic.json && cp $OUT/pwg_layer_traffic.json $R/profiles/pwg_layer_traffic.json
find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
bash $R/tools/r06_bench_only_call.sh $TAG
