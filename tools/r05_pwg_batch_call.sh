#!/bin/bash
# PWG alone by batch size: does the layer kernel get faster per sample when a launch's state fits the 256 MB Infinity Cache?
# usage: tools/r05_pwg_batch_call.sh <tag>
set -u
TAG=${1:-r05r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
{ for B in 1 2 3 4 6 8 16 32; do PK_QPWG_B=$B timeout 100 python tools/quick_pwg.py 2>&1 | grep -E "^PWG|pwg_layer"; done; } > $OUT/pwg_by_batch.txt 2>&1
cat $OUT/pwg_by_batch.txt
