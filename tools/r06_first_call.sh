#!/bin/bash
# Round 6, first gpurun call: the hazard reproducer, the tests added this round, the short bench line.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r06_first_call.sh r06a'      -> gpurun_out/<tag>/
set -u
TAG=${1:-r06a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 300 tools/micro/mfma_chain_hazard 2000 > $OUT/mfma_chain_hazard.txt 2>&1; cat $OUT/mfma_chain_hazard.txt
timeout 1200 python -m pytest tests/test_benchshape_golden_gpu.py tests/test_waveflow_gpu.py tests/test_bench_gpu.py tests/test_pwg_gpu.py -m gpu -q -rA --durations=20 --timeout=600 > $OUT/tests_new.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests_new.log | tail -40
cd /tmp
timeout 400 python $R/bench.py > $OUT/bench.stdout 2> $OUT/bench.err
tail -1 $OUT/bench.stdout > $OUT/bench.json; wc -c $OUT/bench.json; cat $OUT/bench.json; echo
cp $R/profiles/bench_extras_last.json $OUT/bench_extras_last.json 2>/dev/null
tail -5 $OUT/bench.err
ls -la $OUT
