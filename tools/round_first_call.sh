#!/bin/bash
export PK_PROFILE_LIB=1   # the PK_* measurement switches exist in the profile build only: python -m parakeet_amd.build --profile (libpk_synth_prof.so)
# First gpurun call of a round: everything that was added without a GPU gets its first hardware run, then the numbers.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round_first_call.sh r03a'      -> gpurun_out/<tag>/
# 1. the full GPU suite (tests/conftest.py orders the never-run-on-hardware tests last); 2. smoke; 3. the bench line;
# 4. A/B of the TransformerTTS k|v-only prefix projection (PK_TTS_KV_PREFIX); 5. rocprofv3 kernel stats of the bench.
set -u
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=60 --timeout=300 > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -40
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp
timeout 300 python $R/bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
head -c 700 $OUT/bench.json; echo
for kv in 0 1; do
  PK_TTS_KV_PREFIX=$kv timeout 120 python $R/tools/quick_ar.py tts 32 640 > $OUT/quick_tts_kv$kv.log 2>&1
  head -1 $OUT/quick_tts_kv$kv.log
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $f); done
find $OUT -type d -name "stats*" | xargs rm -rf
ls -la $OUT
