#!/bin/bash
# First gpurun call of round 5: the full GPU suite with the tightened bars and the new tests, smoke, the bench line with the new
# extras (text -> wav, exact_f32 in roofline), rocprofv3 kernel stats of the bench.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r05_first_call.sh r05a'      -> gpurun_out/<tag>/
set -u
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -rA --durations=25 --timeout=300 > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -40
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp
timeout 420 python $R/bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench.json
head -c 600 $OUT/bench.json; echo
python - <<PY
import json
try:
    j = json.load(open("$OUT/bench.json"))
    print("value", j["value"], "ms", j["ms_per_step"], "roofline", j["roofline"]["frac"], j["roofline"].get("exact_f32"))
    print(json.dumps(j["extras"].get("text_to_wav"), indent=1)[:3000])
except Exception as e:
    print("bench parse:", e)
PY
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $f); done
find $OUT -type d -name "stats*" | xargs rm -rf
ls -la $OUT
