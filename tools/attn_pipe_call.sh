#!/bin/bash
# A/B of the pipelined attention kernel (PK_FS2_ATTN_PIPE) on the GPU box: FS2 tests with it on, then per-launch durations.
set -u
TAG=${1:-r03attn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
PK_FS2_ATTN_PIPE=1 timeout 600 python -m pytest tests/test_fs2_gpu.py -m gpu -q --timeout=300 -k "ragged or long_utt or alpha or four_heads" 2>&1 | tail -n 2
for p in 0 1 0 1; do
  export PK_FS2_ATTN_PIPE=$p
  bash tools/ffn_planes_prof.sh ${TAG}_$p 0 2>&1 | grep -E "^==|attention"
done
