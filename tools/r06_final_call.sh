#!/bin/bash
# Evidence of round 6 at the final code (run on the GPU box):
#   1. the full GPU suite  2. smoke  3. the bench line as the driver runs it (+ the sidecar)  4. rocprofv3 kernel stats of the bench
#   5. PMC passes: the PWG layer kernel (SQ + traffic -> pwg_layer_traffic.json), the WaveFlow layer kernel (64 / 128 channels, both maths; traffic
#      at 64 channels), the FastSpeech2 kernels.
# usage: tools/r06_final_call.sh <tag>
set -u
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PK_ROUND="round 6"
cd $R
timeout 1800 python -m pytest tests -m gpu -q -rA --durations=25 --timeout=600 > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp
timeout 500 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.stdout 2> $OUT/bench.err
tail -1 $OUT/bench.stdout > $OUT/bench.json; wc -c $OUT/bench.json; cat $OUT/bench.json; echo
cp $R/profiles/bench_extras_last.json $OUT/bench_extras_last.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --extras none > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $f); done
find $OUT -type d -name "stats*" | xargs rm -rf
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"
pmc() { timeout 240 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py $2 > $OUT/pmc_$1.log 2>&1; }
# ---- PWG layer kernel
pmc pA "pwg 32" "$SQ"; pmc pB "pwg 32" "FETCH_SIZE TCC_HIT"; pmc pC "pwg 32" "WRITE_SIZE TCC_MISS TCC_REQ"
python $R/tools/pmc_parse.py $OUT/pmc_pA $OUT/pmc_pB $OUT/pmc_pC --kernel=k_pwg_ > $OUT/pmc_pwg.json
python $R/tools/pmc_traffic.py pwg $OUT/pmc_pwg.json $OUT/pwg_layer_traffic.json; cat $OUT/pwg_layer_traffic.json
# ---- WaveFlow layer kernel
for m in wf wf16 wf128 wf128_16; do
  pmc ${m}_A "$m 8" "$SQ"; pmc ${m}_B "$m 8" "$LDS"
  if [ $m = wf ]; then
    pmc wf_C "wf 8" "FETCH_SIZE TCC_HIT"; pmc wf_D "wf 8" "WRITE_SIZE TCC_MISS TCC_REQ"
    python $R/tools/pmc_parse.py $OUT/pmc_wf_A $OUT/pmc_wf_B $OUT/pmc_wf_C $OUT/pmc_wf_D --kernel=k_wf_ > $OUT/pmc_wf.json
    python $R/tools/pmc_traffic.py wf $OUT/pmc_wf.json $OUT/wf_layer_c64_traffic.json
  else
    python $R/tools/pmc_parse.py $OUT/pmc_${m}_A $OUT/pmc_${m}_B --kernel=k_wf_layer > $OUT/pmc_$m.json
  fi
done
# ---- FastSpeech2
pmc fA "fs2 32" "$SQ"; pmc fB "fs2 32" "$LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"
python $R/tools/pmc_parse.py $OUT/pmc_fA $OUT/pmc_fB --kernel=k_ > $OUT/pmc_fs2.json
find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
ls -la $OUT
# ---- repeat-run identity of the other hot kernels (the WaveFlow gate is in the suite)
cd $R && timeout 400 python tools/repeat_runs.py 40 > $OUT/repeat_runs.txt 2>&1; cat $OUT/repeat_runs.txt
