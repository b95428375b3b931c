#!/bin/bash
export PK_PROFILE_LIB=1   # the PK_* measurement switches exist in the profile build only: python -m parakeet_amd.build --profile (libpk_synth_prof.so)
# WaveFlow layer kernel on the GPU box: tests, per-launch times of the four timed configurations, prefetch A/B, memory ablations, s_memtime trace.
set -u
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_waveflow_gpu.py -m gpu -q -rA --timeout=300 > $OUT/tests.log 2>&1
grep -E "^(PASSED|FAILED|ERROR)|passed|failed|rel err" $OUT/tests.log | tail -12
run() { local tag=$1; shift; env "$@" timeout 200 python tools/quick_wf.py $C $M > $OUT/quick_$tag.log 2>&1; echo "$tag: $(head -1 $OUT/quick_$tag.log | cut -c1-70) | $(grep wf_layer $OUT/quick_$tag.log)"; }
C=64; M=f16x3; run c64 X=1; run c64_noprefetch PK_WF_PREFETCH=0
C=64; M=f16;   run c64_f16 X=1
C=128; M=f16x3; run c128 X=1
C=128; M=f16;   run c128_f16 X=1
for abl in 13 32; do
  PK_WF_ABLATE=$abl timeout 200 python tools/quick_wf_noassert.py 64 > $OUT/quick_abl$abl.log 2>&1; echo "ABL $abl: $(grep wf_layer $OUT/quick_abl$abl.log)"
done
ls $OUT
PK_WF_ABLATE=16 timeout 200 python tools/quick_wf_noassert.py 64 > $OUT/trace.log 2>&1
grep wf_trace $OUT/trace.log | grep "wave [0245] " | tail -8 | cut -c1-230
timeout 300 python -m pytest tests/test_ar_e2e_gpu.py -m gpu -q --timeout=300 2>&1 | tail -n 2
