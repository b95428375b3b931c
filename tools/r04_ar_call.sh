#!/bin/bash
# Round-4 GPU call for the autoregressive models (run on the GPU box): their tests, then the per-kernel engine profile of
# TransformerTTS and Tacotron2 at LJSpeech shape (32 x 640 steps).   usage: tools/r04_ar_call.sh <tag> [pytest -k expression]
set -u
TAG=${1:-r04o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 600 python -m pytest tests/test_tts_gpu.py tests/test_taco2_gpu.py tests/test_ar_benchsize_gpu.py tests/test_ar_e2e_gpu.py -m gpu -q --timeout=300 -x ${2:+-k "$2"} 2>&1 | tail -12) > $OUT/tests.txt
timeout 200 python tools/quick_ar.py tts 32 640 2>&1 | grep -v amdgpu.ids | head -26 > $OUT/quick_tts.txt
timeout 200 python tools/quick_ar.py taco 32 640 2>&1 | grep -v amdgpu.ids | head -18 > $OUT/quick_taco.txt
cat $OUT/tests.txt $OUT/quick_tts.txt $OUT/quick_taco.txt
