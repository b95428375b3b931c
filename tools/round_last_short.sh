#!/bin/bash
# Reduced last call of a round (when the GPU budget is short): the tests that touch Parallel WaveGAN, smoke, the bench line,
# rocprofv3 kernel statistics of the same command.
set -u
TAG=${1:-r03s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 700 python -m pytest tests/test_pwg_gpu.py tests/test_benchshape_gpu.py tests/test_fullsize_gpu.py tests/test_noise_gpu.py tests/test_golden_gpu.py tests/test_speedyspeech_gpu.py -m gpu -q -rA --timeout=300 -k "not waveflow" > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp
timeout 600 python $R/bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
head -c 400 $OUT/bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | while read f; do cp $f $OUT/bench_kernel_stats.csv; done
rm -rf $OUT/stats
