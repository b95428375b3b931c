#!/bin/bash
# One-box A/B of PWG layer kernel variants (tools/build_variant.py -> parakeet_amd/variants/<name>.so copied over the profile library):
#   VARIANTS="a b" bash tools/r06_pwg_ab_call.sh <tag> [reps]
set -u
TAG=${1:-r06r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
{
for rep in $(seq 1 ${2:-3}); do
  timeout 200 python tools/pwg_ab.py product
  for v in ${VARIANTS}; do
    cp parakeet_amd/variants/$v.so parakeet_amd/libpk_synth_prof.so
    PK_PROFILE_LIB=1 timeout 200 python tools/pwg_ab.py $v
  done
done
} 2>&1 | grep -v "amdgpu.ids" | tee $OUT/pwg_ab.txt
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
