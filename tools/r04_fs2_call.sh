#!/bin/bash
# FastSpeech2 tests + timings at batches 1 / 2 / 4 / 16 / 32 (run on the GPU box).  usage: tools/r04_fs2_call.sh <tag>
set -u
TAG=${1:-r04y}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 900 python -m pytest tests/test_fs2_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py tests/test_benchshape_gpu.py tests/test_speedyspeech_gpu.py -m gpu -q --timeout=300 -x 2>&1 | tail -8) > $OUT/tests.txt
{ for B in 1 2 4 16 32; do timeout 100 python tools/quick_fs2.py $B 2>&1 | grep -v amdgpu | head -24; done; } > $OUT/quick_fs2.txt 2>&1
cat $OUT/tests.txt; grep "FS2 B=\|ffn\|attention" $OUT/quick_fs2.txt
