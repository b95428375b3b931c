#!/bin/bash
# PMC pass for the planes feed-forward kernels (run on the GPU box).  usage: tools/pmc_ffnp.sh <tag>
set -u
TAG=${1:-ffnppmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() { timeout 240 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py fs2 32 > $OUT/pmc_$1.log 2>&1; }
pmc A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc B "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"
python $R/tools/pmc_parse.py $OUT/pmc_A $OUT/pmc_B --kernel=k_ > $OUT/pmc_ffnp.json
python - <<PY
import json
d = json.load(open("$OUT/pmc_ffnp.json"))
for k, v in d.items():
    if "ffn" in k or "gemm_h3<2>" in k:
        print(k)
        for kk, vv in sorted(v.items()): print("   ", kk, vv)
        g = v.get("GRBM_GUI_ACTIVE"); ns = v.get("_avg_ns_under_pmc")
        if g and ns: print("    effective clock GHz", g / ns, " mfma busy frac", v["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 4 * 256) if "SQ_VALU_MFMA_BUSY_CYCLES" in v else None)
PY
rm -rf $OUT/pmc_A $OUT/pmc_B
