#!/bin/bash
# Round 5: the WaveFlow determinism fixes -- high-power repeat runs at the benchmark's shape and beyond, the WaveFlow tests, timings.
# usage: tools/r05_wf_fix_call.sh <tag>
set -u
TAG=${1:-r05u}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B8=640,640,640,640,640,640,640,640
{
WF_C=128 WF_FRAMES=$B8 WF_VARIANTS=default,f16 WF_REP=100 timeout 400 python tools/wf_race_bisect.py 2>&1 | grep -v amdgpu
WF_C=128 WF_FRAMES=1700,1700 WF_VARIANTS=default WF_REP=40 timeout 300 python tools/wf_race_bisect.py 2>&1 | grep -v amdgpu
WF_FRAMES=$B8 WF_VARIANTS=default,f16 WF_REP=60 timeout 300 python tools/wf_race_bisect.py 2>&1 | grep -v amdgpu
} > $OUT/determinism.txt 2>&1
cat $OUT/determinism.txt
(timeout 900 python -m pytest tests/test_waveflow_gpu.py tests/test_benchshape_gpu.py tests/test_golden_gpu.py tests/test_released_ckpt_gpu.py -m gpu -q --timeout=600 -k "waveflow" 2>&1 | tail -6) > $OUT/tests.txt
cat $OUT/tests.txt
{ for c in 64 128; do for m in - f16; do timeout 120 python tools/quick_wf.py $c $m 2>&1 | grep -E "^WaveFlow|wf_layer"; done; done; } > $OUT/timing.txt 2>&1
cat $OUT/timing.txt
