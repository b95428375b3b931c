#!/bin/bash
export PK_PROFILE_LIB=1   # the PK_* measurement switches exist in the profile build only: python -m parakeet_amd.build --profile (libpk_synth_prof.so)
# FastSpeech2 on planes (ffn_planes.hip) on the GPU box: tests, error vs the fp64 oracle of both paths, timings per batch size.
set -u
TAG=${1:-r03ffn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_fs2_gpu.py tests/test_fullsize_gpu.py tests/test_benchshape_gpu.py tests/test_tts_gpu.py tests/test_bench_gpu.py -m gpu -q -rA --timeout=300 -k "fs2 or e2e or encoder or bench" > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -12
timeout 300 python tools/ffn_planes_error.py 37 5 64 1 23 > $OUT/error.log 2>&1; tail -2 $OUT/error.log
for B in 32 8 1; do for p in 1 0; do
  echo "B=$B planes=$p: $(PK_FS2_FFN_PLANES=$p timeout 200 python tools/quick_fs2.py $B 2>&1 | grep 'FS2 B')"
done; done
bash tools/ffn_planes_prof.sh $TAG 0
