#!/bin/bash
# Round-end evidence on the GPU box (via gpurun): AR parity tests, smoke, the bench line, rocprofv3 kernel stats of the
# bench program and of the two autoregressive models.   usage: tools/collect_final.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 200 python -m pytest tests/test_tts_gpu.py tests/test_taco2_gpu.py tests/test_noise_gpu.py tests/test_modules_gpu.py -q --timeout=100 > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp
timeout 300 python $R/bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
head -c 600 $OUT/bench.json; echo
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
for m in tts taco; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -o $m -- python $R/tools/quick_ar.py $m 32 640 > $OUT/quick_$m.log 2>&1
  head -12 $OUT/quick_$m.log
done
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $f); done
find $OUT -type d -name "stats*" | xargs rm -rf
ls -la $OUT
