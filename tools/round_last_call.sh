#!/bin/bash
# Last GPU call of a round, on the final commit: the full -m gpu suite, smoke, the bench line, rocprofv3 kernel statistics of
# the same command, PMC passes for the Parallel WaveGAN and the WaveFlow layer kernels.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/round_last_call.sh r03z'      -> gpurun_out/<tag>/
set -u
TAG=${1:-r03z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q -rA --durations=40 --timeout=300 > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp
timeout 600 python $R/bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
head -c 600 $OUT/bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | while read f; do cp $f $OUT/bench_kernel_stats.csv; done
rm -rf $OUT/stats
[ "${2:-}" = nopmc ] && { ls -la $OUT; exit 0; }   # (kernels of the PMC passes unchanged since the last full call)
# ---- PMC: Parallel WaveGAN layer kernel (three passes) -> pwg_layer_traffic.json
pmc() { timeout 240 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py $2 > $OUT/pmc_$1.log 2>&1; }
pmc pA "pwg 32" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc pB "pwg 32" "FETCH_SIZE TCC_HIT"
pmc pC "pwg 32" "WRITE_SIZE TCC_MISS TCC_REQ"
python $R/tools/pmc_parse.py $OUT/pmc_pA $OUT/pmc_pB $OUT/pmc_pC --kernel=k_pwg_ > $OUT/pmc_pwg.json
python $R/tools/pmc_traffic.py pwg $OUT/pmc_pwg.json $OUT/pwg_layer_traffic.json
# ---- PMC: WaveFlow layer kernel
pmc wA "wf 8" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc wB "wf 8" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"
pmc wC "wf 8" "FETCH_SIZE TCC_HIT"
pmc wD "wf 8" "WRITE_SIZE TCC_MISS TCC_REQ"
python $R/tools/pmc_parse.py $OUT/pmc_wA $OUT/pmc_wB $OUT/pmc_wC $OUT/pmc_wD --kernel=k_wf_ > $OUT/pmc_wf.json
python $R/tools/pmc_traffic.py wf $OUT/pmc_wf.json $OUT/wf_layer_c64_traffic.json
find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
ls -la $OUT
