#!/bin/bash
# A/B of the WaveFlow layer kernel's prologue priority (profile build: PK_WF_ABLATE=64 = without) on one box, twice each, both
# maths, 64 and 128 channels; the s_memtime trace of the 12-wave kernel with (16) and without (80).
# usage: tools/r04_wf_prio_call.sh <tag>
set -u
TAG=${1:-r04l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export PK_PROFILE_LIB=1
{
for rep in 1 2; do
for cfg in "64 -" "64 f16" "128 -" "128 f16"; do
  set -- $cfg
  for a in 0 64; do
    echo "PK_WF_ABLATE=$a: $(PK_WF_ABLATE=$a timeout 150 python tools/quick_wf.py $1 $2 0 2>&1 | grep -E "WaveFlow|wf_layer" | tr '\n' '|')"
  done
done
done
} > $OUT/wf_prio_ab.txt 2>&1
for a in 16 80; do
  PK_WF_ABLATE=$a timeout 200 python tools/quick_wf_noassert.py 64 > $OUT/trace_$a.log 2>&1
  grep wf_trace $OUT/trace_$a.log > $OUT/wf_layer_trace_12_waves_abl$a.txt
done
(timeout 400 python -m pytest tests/test_waveflow_gpu.py -m gpu -q --timeout=300 -x 2>&1 | tail -5) > $OUT/tests.txt
cat $OUT/wf_prio_ab.txt; cut -c1-200 $OUT/wf_layer_trace_12_waves_abl16.txt | head -12; cat $OUT/tests.txt
