#!/bin/bash
# PMC passes for the WaveFlow layer kernel (run on the GPU box): SQ / LDS counters and the fabric traffic.
# usage: tools/pmc_wf.sh <tag>      -> gpurun_out/<tag>/pmc_wf.json, wf_layer_c64_traffic.json
set -u
TAG=${1:-wfpmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() { timeout 240 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py wf 8 > $OUT/pmc_$1.log 2>&1; }
pmc A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc B "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"
pmc C "FETCH_SIZE TCC_HIT"
pmc D "WRITE_SIZE TCC_MISS TCC_REQ"
python $R/tools/pmc_parse.py $OUT/pmc_A $OUT/pmc_B $OUT/pmc_C $OUT/pmc_D --kernel=k_wf_ > $OUT/pmc_wf.json
python $R/tools/pmc_traffic.py wf $OUT/pmc_wf.json $OUT/wf_layer_c64_traffic.json
find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
