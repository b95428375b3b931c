#!/bin/bash
# Round-5 GPU call for the 128-channel WaveFlow layer kernel (run on the GPU box):
#   1. the WaveFlow tests (default + bench shape)   2. A/B on one box: product library (operand ring 6 k-steps) against the
#   profile library built with -DPK_WF_RING128=4, both math modes, 2 repetitions interleaved; 64 channels for reference
#   3. counters of the 128-channel kernel (SQ pass), default math and fp16 operands.
# usage: tools/r05_wf_call.sh <tag> [pmc]
set -u
TAG=${1:-r05b}
PMC=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 600 python -m pytest tests/test_waveflow_gpu.py tests/test_benchshape_gpu.py tests/test_golden_gpu.py tests/test_released_ckpt_gpu.py -m gpu -q --timeout=300 -k "waveflow" 2>&1 | tail -15) > $OUT/tests.txt
tail -3 $OUT/tests.txt
{
for rep in 1 2; do
  for cfg in "128 -" "128 f16"; do
    set -- $cfg
    echo "== product (ring ${RINGP:-6})"; timeout 150 python tools/quick_wf.py $1 $2 0 2>&1 | grep -E "WaveFlow|wf_layer"
    echo "== profile library (ring ${RINGQ:-4})"; PK_PROFILE_LIB=1 timeout 150 python tools/quick_wf.py $1 $2 0 2>&1 | grep -E "WaveFlow|wf_layer"
  done
done
for cfg in "64 -" "64 f16"; do
  set -- $cfg
  timeout 150 python tools/quick_wf.py $1 $2 0 2>&1 | grep -E "WaveFlow|wf_layer"
done
} > $OUT/wf_ab.txt 2>&1
cat $OUT/wf_ab.txt
if [ "$PMC" = pmc ]; then
  export TMPDIR=/tmp
  cd /tmp
  for m in wf128 wf128_16; do
    C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
    PK_QWF_PERSISTENT=0 timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${m}_A -o p -- python $R/tools/pmc_run.py $m 8 > $OUT/pmc_${m}_A.log 2>&1
    python $R/tools/pmc_parse.py $OUT/pmc_${m}_A --kernel=k_wf_layer > $OUT/pmc_$m.json
  done
  find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
  python - <<PY
import json
for m in ("wf128", "wf128_16"):
    try:
        j = json.load(open("$OUT/pmc_%s.json" % m))
        print(m, json.dumps(j)[:1500])
    except Exception as e:
        print(m, "pmc parse:", e)
PY
fi
ls -la $OUT
