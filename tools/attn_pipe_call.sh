#!/bin/bash
export PK_PROFILE_LIB=1   # the PK_* measurement switches exist in the profile build only: python -m parakeet_amd.build --profile (libpk_synth_prof.so)
# Attention kernel variants on the GPU box: FS2 tests under each PK_FS2_ATTN_WAVES, then per-launch durations.
set -u
TAG=${1:-r03attn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for w in 5 8; do
  echo "waves $w: $(PK_FS2_ATTN_WAVES=$w timeout 600 python -m pytest tests/test_fs2_gpu.py -m gpu -q --timeout=300 -k 'ragged_batch or long_utt or alpha or four_heads' 2>&1 | tail -n 1)"
done
for w in 4 5 6 8 0; do
  export PK_FS2_ATTN_WAVES=$w
  echo "waves $w"; bash tools/ffn_planes_prof.sh ${TAG}_$w 0 2>&1 | grep -E "^==|attention"
done
