#!/bin/bash
# Round 6: after the wait state in front of the half-wave exchange -- whole inferences against the exact-fp32 path, the configurations that failed in round 5
set -u
TAG=${1:-r06l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export PK_PROFILE_LIB=1 PK_WF_ALLOW_3WAVE=1
echo "== 64 channels, 12-wave workgroups, default math (7 - 25 % of the calls wrong in round 5; 6 of 40 this round before the fix)"
timeout 600 python tools/wf_verify_run.py ${2:-150} 12 2>/dev/null | tail -3 | tee $OUT/c64_w12.txt
echo "== the verifier's instantiation (60 of 60 wrong before the fix)"
PK_WF_ABLATE=128 timeout 600 python tools/wf_verify_run.py 60 12 2>/dev/null | tail -2 | tee $OUT/c64_w12_abl128.txt
echo "== 64 channels, two 6-wave workgroups per CU"
timeout 600 python tools/wf_verify_run.py 60 6 2>/dev/null | tail -2 | tee $OUT/c64_w6.txt
echo "== 128 channels, two working waves per SIMD (1 - 5 % of the calls wrong in round 5)"
WF_C=128 timeout 900 python tools/wf_verify_run.py ${3:-80} 0 2>/dev/null | tail -2 | tee $OUT/c128.txt
echo "== 64 channels, 2 x 2560 frames, 12 waves"
WF_FRAMES=2560,2560 timeout 600 python tools/wf_verify_run.py 60 12 2>/dev/null | tail -2 | tee $OUT/c64_2560.txt
