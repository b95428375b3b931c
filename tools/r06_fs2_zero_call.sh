#!/bin/bash
# Round 6: FastSpeech2 with its per-run clears as one launch per stack / planes buffer (k_zero_list) against one hipMemsetAsync each (variant fs2_old)
set -u
TAG=${1:-r06z1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
for rep in 1 2 3; do for b in 32 16 1; do
  echo -n "product  "; timeout 120 python tools/quick_fs2.py $b 2>&1 | grep -E "^FS2" | tr '\n' ' '; echo
  cp parakeet_amd/variants/fs2_old.so parakeet_amd/libpk_synth_prof.so
  echo -n "fs2_old  "; PK_PROFILE_LIB=1 timeout 120 python tools/quick_fs2.py $b 2>&1 | grep -E "^FS2" | tr '\n' ' '; echo
done; done | tee $OUT/fs2_timings.txt
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
(timeout 1500 python -m pytest tests/test_fs2_gpu.py tests/test_fullsize_gpu.py tests/test_benchshape_gpu.py tests/test_golden_gpu.py tests/test_benchshape_golden_gpu.py tests/test_tts_gpu.py tests/test_speedyspeech_gpu.py tests/test_taco2_gpu.py -m gpu -q --timeout=300 -x 2>&1 | tail -6) | tee $OUT/tests.txt
