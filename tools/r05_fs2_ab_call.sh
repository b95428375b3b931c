#!/bin/bash
# One-box A/B of a FastSpeech2 kernel change (round 5): the product against a variant library (tools/build_variant.py) copied over
# the profile library.  FastSpeech2 tests first.   usage: tools/r05_fs2_ab_call.sh <tag>   (VARIANTS="name ...")
set -u
TAG=${1:-r05m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 700 python -m pytest tests/test_fs2_gpu.py tests/test_fullsize_gpu.py tests/test_benchshape_gpu.py tests/test_golden_gpu.py -m gpu -q --timeout=300 -k "fs2 or fastspeech2 or e2e" 2>&1 | tail -4) > $OUT/tests.txt
tail -2 $OUT/tests.txt
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
run() { timeout 120 python tools/quick_fs2.py $2 2>&1 | grep -E "^FS2|fs2_attention" | tr '\n' ' ' | sed "s/^/$1: /"; echo; }
{
for rep in 1 2 3; do
  for b in 32 16 1; do
    run product $b
    for v in ${VARIANTS:-attn_r04}; do
      cp parakeet_amd/variants/$v.so parakeet_amd/libpk_synth_prof.so
      PK_PROFILE_LIB=1 run $v $b
    done
  done
done
} > $OUT/fs2_ab.txt 2>&1
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
cat $OUT/fs2_ab.txt
