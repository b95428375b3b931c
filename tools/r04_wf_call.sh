#!/bin/bash
# Round-4 GPU call for the WaveFlow row kernel and the row-GEMM / step-attention changes (run on the GPU box):
#   1. the WaveFlow tests + the autoregressive models' tests (row GEMM, step attention)  2. A/B timings: 8 / 12 waves, one launch
#   per layer / per row, both math modes, 64 and 128 channels  3. TransformerTTS / Tacotron2 timings  4. counters of the layer
#   kernel (SQ, LDS passes) for the default and the fp16-operand math.
# usage: tools/r04_wf_call.sh <tag> [pmc]
set -u
TAG=${1:-r04c}
PMC=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 60 tools/micro/grid_barrier > $OUT/grid_barrier.txt 2>&1
(timeout 500 python -m pytest ${TESTS:-tests/test_waveflow_gpu.py} -m gpu -q --timeout=300 2>&1 | tail -15) > $OUT/tests.txt
{
for cfg in "64 - 0 0" "64 - 0 1" "64 - 8 0" "64 f16 0 0" "64 f16 8 0" "128 - 0 0" "128 f16 0 0"; do
  set -- $cfg
  PK_QWF_PERSISTENT=$4 timeout 150 python tools/quick_wf.py $1 $2 $3 2>&1 | grep -E "WaveFlow|wf_layer|wf_row|wf_step"
done
} > $OUT/wf_ab.txt 2>&1
# PWG layer kernel: sensitivity to bytes -- every layer on the FIRST kernel (skip written, not read: 256 of 1 024 B per sample
# and layer gone, x and the gates unchanged; profile build only, the waveform is wrong)
{ for a in 0 1024 0 1024; do echo "PK_PWG_ABLATE=$a: $(PK_PROFILE_LIB=1 PK_PWG_ABLATE=$a timeout 150 python tools/quick_pwg.py 2>&1 | grep -E 'PWG B|pwg_layer_h3' | tr '\n' '|')"; done; } > $OUT/pwg_no_skip_read.txt 2>&1
timeout 200 python tools/quick_ar.py tts 32 640 2>&1 | head -24 > $OUT/quick_tts.txt
timeout 200 python tools/quick_ar.py taco 32 640 2>&1 | head -16 > $OUT/quick_taco.txt
if [ "$PMC" = pmc ]; then
  export TMPDIR=/tmp
  cd /tmp
  for m in wf wf16; do
    for p in A B; do
      if [ $p = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"; fi
      PK_QWF_PERSISTENT=0 timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${m}_$p -o p -- python $R/tools/pmc_run.py $m 8 > $OUT/pmc_${m}_$p.log 2>&1
    done
    python $R/tools/pmc_parse.py $OUT/pmc_${m}_A $OUT/pmc_${m}_B --kernel=k_wf_layer > $OUT/pmc_$m.json
  done
  find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
  python - <<PY
import json
for m in ("wf","wf16"):
    d=json.load(open("$OUT/pmc_%s.json"%m))
    for k,v in d.items():
        if ", 3, 0" in k:
            print(m, k, "us", round(v["_avg_ns_under_pmc"]/1e3,1), "mfma_busy", round(v["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*v["GRBM_GUI_ACTIVE"]/8),3), "valu_qc_per_mfma", round(v["SQ_ACTIVE_INST_VALU"]/v["SQ_INSTS_MFMA"],2), "wait_any", round(v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"],3), "wait_inst_any", round(v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"],3), "clk_GHz", round(v["GRBM_GUI_ACTIVE"]/8/v["_avg_ns_under_pmc"],3), "lds_conflict", round(v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_LDS_IDX_ACTIVE",1),1),3))
PY
fi > $OUT/pmc_summary.txt 2>&1
cat $OUT/grid_barrier.txt $OUT/tests.txt $OUT/wf_ab.txt $OUT/pwg_no_skip_read.txt $OUT/quick_tts.txt $OUT/quick_taco.txt $OUT/pmc_summary.txt
