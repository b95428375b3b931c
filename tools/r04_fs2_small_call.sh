#!/bin/bash
# FastSpeech2 at small batches: the tiling options, one box.  usage: tools/r04_fs2_small_call.sh <tag>
set -u
TAG=${1:-r04x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
{
for B in 1 2 4 8 16; do
for o in "ffn_one_tile_max=0" "" "ffn_one_tile_max=1024" "ffn_one_tile_max=2048" "ffn_one_tile_max=4096" "ffn_one_tile_max=1024,attn_waves=4"; do
  PK_QFS2_OPTS=$o timeout 100 python tools/quick_fs2.py $B 2>&1 | grep "FS2 B="
done
done
} > $OUT/fs2_small_batch_options.txt 2>&1
PK_QFS2_OPTS=ffn_one_tile_max=1024 timeout 100 python tools/quick_fs2.py 1 2>&1 | grep -v amdgpu > $OUT/fs2_b1_one_tile_1024.txt
cat $OUT/fs2_small_batch_options.txt; cat $OUT/fs2_b1_one_tile_1024.txt
