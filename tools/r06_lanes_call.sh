#!/bin/bash
# Round 6: the two-lane issue order in the product (Synthesizer.add_acoustic_lane, bench --pipeline 2) -- test, then the bench with 2 / 1 alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06q}; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_bench_gpu.py -m gpu -q --timeout=500 2>&1 | tail -4 | tee $OUT/tests.txt
cd /tmp
for rep in 1 2; do for p in 2 1; do
  timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras none --pipeline $p 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipeline $p:', d['value'], d['ms_per_step'], d['pipeline_check'], d['config']['pipeline'][:60])"
done; done | tee $OUT/bench_lanes.txt
