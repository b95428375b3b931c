#!/bin/bash
# FastSpeech2 at batch 1 (one 128-token utterance -> 640 frames): the engine's per-kernel profile and rocprofv3 kernel durations.
# usage: tools/r04_fs2_b1_call.sh <tag>
set -u
TAG=${1:-r04w}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for B in 1 4; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$B -o p -- python $R/tools/quick_fs2.py $B > $OUT/quick_fs2_$B.txt 2>&1
  f=$(find $OUT/kt_$B -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/fs2_kernel_stats_b$B.csv
  rm -rf $OUT/kt_$B
done
for B in 1 4; do grep -v "amdgpu.ids\|rocprofv3\|HSA version" $OUT/quick_fs2_$B.txt | head -30; head -22 $OUT/fs2_kernel_stats_b$B.csv | cut -c1-200; done
