#!/bin/bash
# Final evidence of round 5 at the final code (run on the GPU box):
#   1. the full GPU suite  2. smoke  3. the bench line  4. rocprofv3 kernel stats of the bench  5. SQ counters of the WaveFlow layer
#   kernel: 64 / 128 channels, default math / fp16 operands.
# usage: tools/r05_final_call.sh <tag>
set -u
TAG=${1:-r05z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -rA --durations=15 --timeout=300 > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp
timeout 420 python $R/bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python - <<PY
import json
try:
    j = json.load(open("$OUT/bench.json"))
    e = j["extras"]
    print("value", j["value"], "ms", j["ms_per_step"], "roofline", j["roofline"]["frac"], "unpipelined", j.get("value_unpipelined"))
    for k in ("waveflow_c64_batch8", "waveflow_c64_batch8_fp16", "waveflow_c128_batch8", "waveflow_c128_batch8_fp16"):
        print(k, round(e[k]["ms_per_batch"], 2), e[k].get("roofline", {}).get("avg_launch_ms"))
    for k in ("fastspeech2_batch32", "fastspeech2_batch16", "fastspeech2_batch1"):
        print(k, e[k]["ms_per_batch"])
except Exception as ex:
    print("bench parse:", ex)
PY
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $f); done
find $OUT -type d -name "stats*" | xargs rm -rf
for m in wf wf16 wf128 wf128_16; do
  C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
  PK_QWF_PERSISTENT=0 timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${m}_A -o p -- python $R/tools/pmc_run.py $m 8 > $OUT/pmc_${m}_A.log 2>&1
  C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"
  PK_QWF_PERSISTENT=0 timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${m}_B -o p -- python $R/tools/pmc_run.py $m 8 > $OUT/pmc_${m}_B.log 2>&1
  python $R/tools/pmc_parse.py $OUT/pmc_${m}_A $OUT/pmc_${m}_B --kernel=k_wf_layer > $OUT/pmc_$m.json
done
find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
ls -la $OUT
