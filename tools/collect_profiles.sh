#!/bin/bash
# Run on the GPU box (via gpurun): bench JSON, rocprofv3 kernel stats, PMC passes for the PWG kernels.
# usage: tools/collect_profiles.sh <tag>     outputs under gpurun_out/<tag>/
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 python $R/bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
pmc() { timeout 240 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$1 -o p -- python $R/tools/pmc_run.py pwg 32 > $OUT/pmc_$1.log 2>&1; }
pmc A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc B "FETCH_SIZE TCC_HIT"
pmc C "WRITE_SIZE TCC_MISS TCC_REQ"
python $R/tools/pmc_parse.py $OUT/pmc_A $OUT/pmc_B $OUT/pmc_C --kernel=k_pwg_ > $OUT/pmc_pwg.json
cat $OUT/bench.json | head -c 1500; echo
python - <<PY
import json
d = json.load(open("$OUT/pmc_pwg.json"))
LK = [k for k in d if k.startswith("k_pwg_layer") and "false" in k][0]
L = d[LK]; F = d["k_pwg_first"]; Z = d.get("k_pwg_last_h3") or d["k_pwg_last"]
n = 32 * 163840
wcal = F["WRITE_SIZE"] * 1024 / (64 * 4 * n)      # known bytes: 64 channels x 4 B per sample written
rcal = Z["FETCH_SIZE"] * 1024 / (64 * 4 * n)      # known bytes: 64 channels x 4 B per sample read (same dword-per-lane pattern)
hbm = L["FETCH_SIZE"] * 1024 / rcal + L["WRITE_SIZE"] * 1024 / wcal
clk = L["GRBM_GUI_ACTIVE"] / 8 / (L["_avg_ns_under_pmc"] * 1e-9)
targs = [t.strip() for t in LK[LK.index("<") + 1:LK.rindex(">")].split(",")]   # <FIRST, HALF[, ABL]>
prof_key = ("pwg_layer_h3" if targs[1] == "true" else "pwg_layer_b3") if "b3" in LK else "pwg_layer"
out = {"kernel": LK, "prof_key": prof_key, "hbm_bytes_per_launch": hbm,
       "fetch_size_kb": L["FETCH_SIZE"], "write_size_kb": L["WRITE_SIZE"],
       "fetch_calibration": rcal, "write_calibration": wcal,
       "calibration_note": "FETCH_SIZE calibrated on k_pwg_last[_h3] (reads exactly 64x4 B/sample with the same dword-per-lane, 128-B-segment pattern), WRITE_SIZE on k_pwg_first (writes exactly 64x4 B/sample); MI355X_MICROARCH.md HBM section: FETCH_SIZE under-counts wide streams by 2x on gfx950",
       "mfma_busy_frac": L["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * L["GRBM_GUI_ACTIVE"] / 8),
       "effective_clock_ghz": clk / 1e9,
       "wait_any_frac": L["SQ_WAIT_ANY"] / L["SQ_WAVE_CYCLES"],
       "wait_inst_any_frac": L["SQ_WAIT_INST_ANY"] / L["SQ_WAVE_CYCLES"],
       "l2_hit_rate": L["TCC_HIT"] / (L["TCC_HIT"] + L["TCC_MISS"]),
       "samples_per_launch": n}
json.dump(out, open("$OUT/pwg_layer_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
