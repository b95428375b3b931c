#!/bin/bash
# Round 6: the scale guard's maxima from the layer kernel's epilogue (AMAX instantiations) against the 30 separate passes (variant guard_old = HEAD~).
set -u
TAG=${1:-r06g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
{
for rep in 1 2; do
  timeout 300 python tools/pwg_guard_cost.py product
  cp parakeet_amd/variants/guard_old.so parakeet_amd/libpk_synth_prof.so
  PK_PROFILE_LIB=1 timeout 300 python tools/pwg_guard_cost.py guard_old
  timeout 200 python tools/pwg_ab.py product
  PK_PROFILE_LIB=1 timeout 200 python tools/pwg_ab.py guard_old
done
} 2>&1 | grep -v "amdgpu.ids" | tee $OUT/guard_cost.txt
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
timeout 900 python -m pytest tests -m gpu -q -k "pwg or wavegan or e2e or synth or benchshape" --timeout=600 2>&1 | tail -5 | tee $OUT/tests.txt
