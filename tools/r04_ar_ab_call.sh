#!/bin/bash
# TransformerTTS decode at LJSpeech shape: the side-stream options, one box.  usage: tools/r04_ar_ab_call.sh <tag>
set -u
TAG=${1:-r04t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
{
for rep in 1 2; do
for o in "" "overlap_cu_mask=1" "overlap_cu_mask=0" "overlap_prefix=0"; do
  PK_QAR_OPTS=$o timeout 200 python tools/quick_ar.py tts 32 640 2>&1 | grep "tts B="
done
done
timeout 200 python tools/quick_ar.py taco 32 640 2>&1 | grep "taco B="
} > $OUT/tts_options_ab.txt 2>&1
cat $OUT/tts_options_ab.txt
