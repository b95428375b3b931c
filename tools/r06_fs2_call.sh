#!/bin/bash
# Round 6: FastSpeech2 after the attention kernel's vector diet: the FS2 tests, then timings at 32 / 16 / 1 utterances (three repetitions)
set -u
TAG=${1:-r06s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 900 python -m pytest tests/test_fs2_gpu.py tests/test_fullsize_gpu.py tests/test_benchshape_gpu.py tests/test_golden_gpu.py tests/test_benchshape_golden_gpu.py tests/test_tts_gpu.py tests/test_speedyspeech_gpu.py -m gpu -q --timeout=300 -x 2>&1 | tail -6) | tee $OUT/tests.txt
for rep in 1 2 3; do for b in 32 16 1; do timeout 120 python tools/quick_fs2.py $b 2>&1 | grep -E "^FS2|fs2_attention" | tr '\n' ' '; echo; done; done | tee $OUT/fs2_timings.txt
