#!/bin/bash
# rocprofv3 kernel durations of the TransformerTTS decode (LJSpeech shape) next to the engine's HIP-event profile: 640 steps and
# 64 steps (short prefix and key / value caches: what is left when the weights are the only traffic).
# usage: tools/r04_ar_trace_call.sh <tag>
set -u
TAG=${1:-r04s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for L in 640 64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$L -o p -- python $R/tools/quick_ar.py tts 32 $L > $OUT/quick_tts_$L.txt 2>&1
  f=$(find $OUT/kt_$L -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/tts_kernel_stats_$L.csv
  rm -rf $OUT/kt_$L
done
cd $R
for L in 640 64; do grep -v amdgpu.ids $OUT/quick_tts_$L.txt | head -20; head -12 $OUT/tts_kernel_stats_$L.csv | cut -c1-170; done
