#!/bin/bash
# s_memtime trace of the 12-wave 64-channel layer kernel, default math and fp16 operands (profile library, PK_WF_ABLATE=16):
# stamps of every wave of workgroup 5 in the last traced launch -- 0 round start, 1 / 2 before / after the prologue barrier,
# 3 + 2g / 4 + 2g before / after the barrier that ends slab g, 19 gate start, 20 gate done, 21 out projection done, 22 stores, 23 end.
set -u
TAG=${1:-r05e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for m in - f16; do
  PK_PROFILE_LIB=1 PK_WF_ABLATE=16 timeout 200 python tools/quick_wf.py 64 $m 0 > $OUT/trace_$m.log 2>&1
  grep -E "WaveFlow|wf_layer " $OUT/trace_$m.log
  grep wf_trace $OUT/trace_$m.log | tail -12
done
