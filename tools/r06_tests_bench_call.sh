#!/bin/bash
# Round 6: the full GPU suite, smoke, the bench line (+ sidecar) and its rocprofv3 kernel stats
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r06_tests_bench_call.sh r06m'
set -u
TAG=${1:-r06m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1800 python -m pytest tests -m gpu -q -rA --durations=25 --timeout=600 > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -30
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp
timeout 400 python $R/bench.py > $OUT/bench.stdout 2> $OUT/bench.err
tail -1 $OUT/bench.stdout > $OUT/bench.json; wc -c $OUT/bench.json; cat $OUT/bench.json; echo
cp $R/profiles/bench_extras_last.json $OUT/bench_extras_last.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --extras none > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $f); done
find $OUT -type d -name "stats*" | xargs rm -rf
ls $OUT
