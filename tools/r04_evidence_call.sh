#!/bin/bash
# Round-4 evidence call (run on the GPU box): 1. the barrier microbenchmark tools/micro/grid_barrier2  2. the s_memtime trace of the
# 12-wave WaveFlow layer kernel (profile build)  3. rocprofv3 kernel durations of the WaveFlow layer kernel next to the HIP-event
# figures of the same run  4. SQ / LDS counters of the FastSpeech2 kernels (the 256-channel convs on planes included).
# usage: tools/r04_evidence_call.sh <tag>
set -u
TAG=${1:-r04k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 150 tools/micro/grid_barrier2 > $OUT/grid_barrier2.txt 2>&1
echo "exit $?" >> $OUT/grid_barrier2.txt
PK_PROFILE_LIB=1 PK_WF_ABLATE=16 timeout 200 python tools/quick_wf_noassert.py 64 > $OUT/trace.log 2>&1
grep wf_trace $OUT/trace.log > $OUT/wf_layer_trace_12_waves.txt
export TMPDIR=/tmp
cd /tmp
for m in - f16; do
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_wf_$m -o p -- python $R/tools/quick_wf.py 64 $m 0 > $OUT/kt_wf_$m.log 2>&1
  f=$(find $OUT/kt_wf_$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/wf_kernel_stats_$m.csv
  rm -rf $OUT/kt_wf_$m
done
bash $R/tools/pmc_fs2.sh $TAG/fs2 > $OUT/pmc_fs2_summary.txt 2>&1
cd $R
cat $OUT/grid_barrier2.txt; head -12 $OUT/wf_layer_trace_12_waves.txt | cut -c1-250; grep -E "WaveFlow|wf_layer" $OUT/kt_wf_-.log $OUT/kt_wf_f16.log; head -4 $OUT/wf_kernel_stats_-.csv $OUT/wf_kernel_stats_f16.csv
