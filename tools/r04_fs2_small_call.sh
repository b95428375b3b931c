#!/bin/bash
# FastSpeech2 at small batches: the tiling options, one box.  usage: tools/r04_fs2_small_call.sh <tag>
set -u
TAG=${1:-r04x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
{
for B in 1 2 4 8; do
for o in "" "ffn_planes=0" "ffnp_variant=44" "ffnp_variant=48" "ffnp_variant=84" "attn_waves=4" "attn_waves=8"; do
  PK_QFS2_OPTS=$o timeout 100 python tools/quick_fs2.py $B 2>&1 | grep "FS2 B="
done
done
} > $OUT/fs2_small_batch_options.txt 2>&1
PK_QFS2_OPTS=ffn_planes=0 timeout 100 python tools/quick_fs2.py 1 2>&1 | grep -v amdgpu > $OUT/fs2_b1_tile_path.txt
cat $OUT/fs2_small_batch_options.txt; cat $OUT/fs2_b1_tile_path.txt
