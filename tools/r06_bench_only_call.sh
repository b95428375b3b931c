#!/bin/bash
# The bench as the driver runs it (line + sidecar) and its rocprofv3 kernel stats, with profiles/pwg_layer_traffic.json already collected on this kernel source.
set -u
TAG=${1:-r06b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PK_ROUND="round 6"
cd /tmp
timeout 500 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.stdout 2> $OUT/bench.err
tail -1 $OUT/bench.stdout > $OUT/bench.json; wc -c $OUT/bench.json; cat $OUT/bench.json; echo
cp $R/profiles/bench_extras_last.json $OUT/bench_extras_last.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --extras none > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $f); done
find $OUT -type d -name "stats*" | xargs rm -rf
