#!/bin/bash
# Round 6: the 12-wave default-math kernel under the idle wave's LDS slab verifier (tools/wf_verify_run.py)
set -u
TAG=${1:-r06d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export PK_PROFILE_LIB=1 PK_WF_ALLOW_3WAVE=1
timeout 300 python tools/wf_verify_run.py ${2:-40} 12 > $OUT/plain.txt 2> $OUT/plain.err; tail -12 $OUT/plain.txt
PK_WF_ABLATE=128 timeout 400 python tools/wf_verify_run.py ${3:-60} 12 > $OUT/verify.txt 2> $OUT/verify.err; tail -12 $OUT/verify.txt
PK_WF_ABLATE=128 PK_WF_VERIFY_OFF=1 timeout 400 python tools/wf_verify_run.py ${3:-60} 12 > $OUT/verify_off.txt 2> $OUT/verify_off.err; tail -4 $OUT/verify_off.txt
grep -c "wf_verify: 0 LDS" $OUT/verify.err; grep "wf_verify" $OUT/verify.err | grep -v ": 0 LDS" | head -60
