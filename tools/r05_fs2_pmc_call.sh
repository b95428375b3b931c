#!/bin/bash
# FastSpeech2 at the final code of round 5: per-kernel timings at batches 32 / 16 / 1 and the SQ counter passes (tools/pmc_fs2.sh).
# usage: tools/r05_fs2_pmc_call.sh <tag>
set -u
TAG=${1:-r05q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
{ for B in 32 16 1; do timeout 100 python tools/quick_fs2.py $B 2>&1 | grep -v amdgpu | head -26; done; } > $OUT/quick_fs2.txt 2>&1
grep "FS2 B=\|ffn\|attention\|layernorm" $OUT/quick_fs2.txt
bash tools/pmc_fs2.sh $TAG 2>&1 | tail -40
ls -la $OUT
