#!/bin/bash
# Round 6: the PWG layer kernel per dilation -- duration, FETCH_SIZE / TCC_HIT, WRITE_SIZE / TCC_MISS / TCC_REQ of each of the 30 launches of a call.
set -u
TAG=${1:-r06d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() { timeout 240 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py pwg 32 > $OUT/pmc_$1.log 2>&1; }
pmc B "FETCH_SIZE TCC_HIT"; pmc C "WRITE_SIZE TCC_MISS TCC_REQ"; pmc E "TCC_EA_RDREQ TCC_EA_RDREQ_32B"; pmc A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
python $R/tools/pmc_per_launch.py $OUT/pmc_B $OUT/pmc_C $OUT/pmc_E $OUT/pmc_A --kernel=k_pwg_layer > $OUT/per_launch.txt 2>&1
cat $OUT/per_launch.txt
# durations without counters: kernel trace of three calls
timeout 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o k -- python $R/tools/pmc_run.py pwg 32 > $OUT/kt.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
r = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) for x in csv.DictReader(open(f)) if "k_pwg_layer" in x["Kernel_Name"])
print("plain kernel trace, us per layer launch in order:", [round(d / 1000) for _, d in r])
PY
find $OUT -maxdepth 1 -type d \( -name "pmc_*" -o -name kt \) | xargs rm -rf
