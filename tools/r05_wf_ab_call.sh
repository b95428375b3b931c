#!/bin/bash
# One-box A/B of the WaveFlow layer kernel (round 5): the product against variant libraries (tools/build_variant.py) copied over
# the profile library -- r04 = wf_layer.hip of the round-4 final code; wst64 / ahead2 / wst64_ahead2 = experiment switches.
# usage: tools/r05_wf_ab_call.sh <tag>
set -u
TAG=${1:-r05c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 600 python -m pytest tests/test_waveflow_gpu.py tests/test_benchshape_gpu.py tests/test_golden_gpu.py -m gpu -q --timeout=300 -k "waveflow" 2>&1 | tail -6) > $OUT/tests.txt
tail -2 $OUT/tests.txt
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
run() {   # run <label> <C> <math>
  timeout 150 python tools/quick_wf.py $2 $3 0 2>&1 | grep -E "WaveFlow|wf_layer" | tr '\n' ' ' | sed "s/^/$1: /"; echo
}
{
for rep in 1 2; do
  for cfg in "128 -" "128 f16" "64 -" "64 f16"; do
    set -- $cfg
    run product $1 $2
    for v in ${VARIANTS:-r04 linw0 big16}; do
      [ "$1" = 64 ] && [ $v = big16 ] && continue          # (a 128-channel switch)
      [ -f parakeet_amd/variants/$v.so ] || continue
      cp parakeet_amd/variants/$v.so parakeet_amd/libpk_synth_prof.so
      PK_PROFILE_LIB=1 run $v $1 $2
    done
  done
done
} > $OUT/wf_ab.txt 2>&1
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
cat $OUT/wf_ab.txt
