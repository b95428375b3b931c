#!/bin/bash
# One-box A/B of WaveFlow layer kernel variants (tools/build_variant.py -> parakeet_amd/variants/<name>.so copied over the profile library)
#   VARIANTS="a b" bash tools/r06_wf_ab_call.sh <tag> [reps]
set -u
TAG=${1:-r06t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
run() { timeout 150 python tools/quick_wf.py $2 $3 0 2>&1 | grep -E "WaveFlow|wf_layer" | tr '\n' ' ' | sed "s/^/$1: /; s/Msamples.*host enqueue [0-9.]* ms.batch//"; echo; }
{
for rep in $(seq 1 ${2:-2}); do
  for cfg in ${CONFIGS:-"64:-" "64:f16" "128:-" "128:f16"}; do
    c=${cfg%%:*}; m=${cfg##*:}
    run product $c $m
    for v in ${VARIANTS}; do
      cp parakeet_amd/variants/$v.so parakeet_amd/libpk_synth_prof.so
      PK_PROFILE_LIB=1 run $v $c $m
    done
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/wf_ab.txt
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
