#!/bin/bash
# Round 3, second GPU call: first hardware run of the plane-format WaveFlow kernel + the tests the first call did not reach.
set -u
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_waveflow_gpu.py tests/test_ar_e2e_gpu.py tests/test_bench_gpu.py tests/test_ar_benchsize_gpu.py \
    "tests/test_checkpoint_gpu.py::test_example_recipe_script_raw_text" -m gpu -q -rA --durations=20 --timeout=300 > $OUT/tests_a.log 2>&1
grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $OUT/tests_a.log | tail -30
timeout 600 python -m pytest tests/test_tts_gpu.py -m gpu -q -rA --timeout=300 -k "dec_concat or all_post_concat or speaker_embeddings or reduction_factor or style_tokens or kv_only" > $OUT/tests_b.log 2>&1
grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $OUT/tests_b.log | tail -20
for c in 64 128; do
  timeout 200 python tools/quick_wf.py $c > $OUT/quick_wf_$c.log 2>&1; head -1 $OUT/quick_wf_$c.log; grep wf_layer $OUT/quick_wf_$c.log
done
PK_WF_ACTIVE=4 timeout 200 python tools/quick_wf.py 64 > $OUT/quick_wf_64_active4.log 2>&1; head -1 $OUT/quick_wf_64_active4.log
ls -la $OUT
