#!/bin/bash
# the last GPU call of round 6: what a guarded / a sampled PWG call costs, then counters + bench (tools/r06_traffic_bench_call.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/$1; cd $R
for rep in 1 2; do timeout 300 python tools/pwg_guard_cost.py product; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$1/guard_cost.txt
bash tools/r06_traffic_bench_call.sh $1
