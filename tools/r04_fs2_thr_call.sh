#!/bin/bash
# FastSpeech2 at batches 16 / 32: the one-tile threshold, one box.  usage: tools/r04_fs2_thr_call.sh <tag>
set -u
TAG=${1:-r04aa}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
{
for rep in 1 2; do
for B in 16 32; do
for o in "ffn_one_tile_max=0" "" "ffn_one_tile_max=4096" "ffn_one_tile_max=8192" "ffn_one_tile_max=32768"; do
  PK_QFS2_OPTS=$o timeout 100 python tools/quick_fs2.py $B 2>&1 | grep "FS2 B="
done
done
done
} > $OUT/fs2_one_tile_threshold.txt 2>&1
cat $OUT/fs2_one_tile_threshold.txt
