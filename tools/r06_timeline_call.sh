#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06tl}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --extras none > $OUT/kt.log 2>&1
python $R/tools/step_timeline.py $OUT/kt | tee $OUT/timeline.txt
rm -rf $OUT/kt
