#!/bin/bash
# Where a WaveFlow batch's wall time goes beyond its kernels: host enqueue time (tools/quick_wf.py) and the device-side gaps
# between consecutive kernels (rocprofv3 --kernel-trace timestamps, tools/kernel_gaps.py), 64 channels, both maths.
set -u
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for m in - f16; do timeout 150 python tools/quick_wf.py 64 $m 0 2>&1 | grep -E "WaveFlow|wf_layer "; done | tee $OUT/quick.txt
cd /tmp
for m in - f16; do
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$m -o k -- python $R/tools/quick_wf.py 64 $m 0 > $OUT/kt_$m.log 2>&1
  f=$(find $OUT/kt_$m -name "*kernel_trace.csv" | head -1)
  echo "== math $m: $f"; python $R/tools/kernel_gaps.py $f k_wf | tee $OUT/gaps_$m.txt
done
find $OUT -maxdepth 1 -type d -name "kt_*" | xargs rm -rf
