#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for m in wf16 wf128 wf128_16; do
  for p in A B; do
    if [ $p = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"; fi
    timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${m}_$p -o p -- python $R/tools/pmc_run.py $m 8 > $OUT/pmc_${m}_$p.log 2>&1
  done
  python $R/tools/pmc_parse.py $OUT/pmc_${m}_A $OUT/pmc_${m}_B --kernel=k_wf_layer > $OUT/pmc_$m.json
done
find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
python - <<PY
import json
for m in ("wf16","wf128","wf128_16"):
    d=json.load(open("$OUT/pmc_%s.json"%m))
    for k,v in d.items():
        if ", 3, 0" in k:
            print(m, k, "us", round(v["_avg_ns_under_pmc"]/1e3,1), "mfma_busy", round(v["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*v["GRBM_GUI_ACTIVE"]/8),3), "valu_qc_per_mfma", round(v["SQ_ACTIVE_INST_VALU"]/v["SQ_INSTS_MFMA"],2), "wait_any", round(v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"],3), "clk", round(v["GRBM_GUI_ACTIVE"]/8/v["_avg_ns_under_pmc"],3))
PY
