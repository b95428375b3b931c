#!/bin/bash
# Round 6: replay one layer launch (a flow's first layer, row 5: 9 taps) many times against the 8-wave kernel's result (waveflow.hip PK_WF_REPLAY)
#   bash tools/r06_wf_replay_call.sh <tag> <repeats with the verifier instantiation> <repeats of the plain 12-wave kernel>
set -u
TAG=${1:-r06g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export PK_PROFILE_LIB=1 PK_WF_ALLOW_3WAVE=1
PK_WF_ABLATE=128 PK_WF_REPLAY=32:${2:-3000} timeout 600 python tools/wf_verify_run.py 1 12 > $OUT/v_on.txt 2> $OUT/v_on.err; grep "wf_replay" $OUT/v_on.err | grep -v "   pos" | tail -12
PK_WF_REPLAY=32:${3:-30000} timeout 600 python tools/wf_verify_run.py 1 12 > $OUT/plain.txt 2> $OUT/plain.err; grep "wf_replay" $OUT/plain.err | grep -v "   pos" | tail -8
