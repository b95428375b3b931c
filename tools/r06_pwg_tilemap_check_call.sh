#!/bin/bash
# Round 6: the adopted tile map (PK_PWG_TILE_MAP=1, the product) against workgroup-major (variant tilemap0) at 1 / 4 / 32 utterances, then the PWG GPU tests.
set -u
TAG=${1:-r06u}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
{
for B in 1 4 32; do
  echo "-- $B utterances x 640 frames"
  for rep in 1 2 3; do
    PK_QPWG_B=$B timeout 200 python tools/pwg_ab.py product
    cp parakeet_amd/variants/tilemap0.so parakeet_amd/libpk_synth_prof.so
    PK_QPWG_B=$B PK_PROFILE_LIB=1 timeout 200 python tools/pwg_ab.py tilemap0
  done
done
} 2>&1 | grep -v "amdgpu.ids" | tee $OUT/pwg_ab.txt
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
timeout 900 python -m pytest tests -m gpu -q -k "pwg or wavegan or e2e or synth or benchshape" --timeout=600 2>&1 | tail -5 | tee $OUT/tests.txt
