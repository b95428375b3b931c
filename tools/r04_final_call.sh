#!/bin/bash
# Round-4 end-of-round evidence on the GPU box (via gpurun): the full -m gpu suite, smoke(), the bench line, rocprofv3 kernel
# stats of the bench program and of TransformerTTS, the PMC traffic passes of the PWG and WaveFlow layer kernels.
# usage: tools/r04_final_call.sh <tag> [pmc]   -> gpurun_out/<tag>/
set -u
TAG=${1:-r04z}
PMC=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
(timeout 800 python -m pytest tests -m gpu -q --timeout=300 2>&1 | tail -25) > $OUT/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
cd /tmp
timeout 300 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_tts -o tts -- python $R/tools/quick_ar.py tts 32 640 > $OUT/quick_tts.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $f); done
find $OUT -type d -name "stats*" | xargs rm -rf
if [ "$PMC" = pmc ]; then
  pmc() { timeout 120 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py $2 > $OUT/pmc_$1.log 2>&1; }
  pmc pA "pwg 32" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
  pmc pB "pwg 32" "FETCH_SIZE TCC_HIT"
  pmc pC "pwg 32" "WRITE_SIZE TCC_MISS TCC_REQ"
  python $R/tools/pmc_parse.py $OUT/pmc_pA $OUT/pmc_pB $OUT/pmc_pC --kernel=k_pwg_ > $OUT/pmc_pwg.json
  python $R/tools/pmc_traffic.py pwg $OUT/pmc_pwg.json $OUT/pwg_layer_traffic.json
  pmc wC "wf 8" "FETCH_SIZE TCC_HIT"
  pmc wD "wf 8" "WRITE_SIZE TCC_MISS TCC_REQ"
  python $R/tools/pmc_parse.py $OUT/pmc_wC $OUT/pmc_wD --kernel=k_wf_ > $OUT/pmc_wf_traffic.json
  python $R/tools/pmc_traffic.py wf $OUT/pmc_wf_traffic.json $OUT/wf_layer_c64_traffic.json
  find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
fi
tail -4 $OUT/gputest.txt; tail -1 $OUT/smoke.log; head -c 500 $OUT/bench.json; echo; ls $OUT
