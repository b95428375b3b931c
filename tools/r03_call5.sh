#!/bin/bash
set -u
TAG=${1:-r03f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
PK_WF_ABLATE=16 timeout 200 python tools/quick_wf_noassert.py 64 > $OUT/trace.log 2>&1
grep wf_trace $OUT/trace.log | tail -16
grep wf_layer $OUT/trace.log
