"""HBM traffic per launch of the two layer kernels from the PMC passes (tools/round_last_call.sh).
usage: pmc_traffic.py pwg|wf <pmc json of tools/pmc_parse.py> <out json>

Counter handling as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE / WRITE_SIZE are KB of L2 <-> fabric requests
(Infinity Cache hits included); on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, other patterns are
uncalibrated -- so every factor is taken from a kernel of the same run whose byte count is known exactly:
  pwg:  reads  k_pwg_last[_h3]   (64 ch x 4 B per sample, dword per lane -- the layer kernel's pattern)
        writes k_pwg_first       (64 ch x 4 B per sample); without it (noise-fed first block) the first block's own launch (2 x 64 ch x 4 B)
  wf:   reads  the guide's factor 2 (16 B per lane, the layer kernel's only access width), cross-checked on
               k_wf_cond_planes (reads and rewrites 96 x 32 x 4 B per block)
        writes k_wf_cond_planes
"""
import json
import sys

kind, src, dst = sys.argv[1:4]
d = json.load(open(src))
if kind == "pwg":
    LK = ([k for k in d if k.startswith("k_pwg_layer_b3<false") or k.startswith("k_pwg_layer<false")] or
          [k for k in d if k.startswith("k_pwg_layer") and "false" in k])[0]
    def pick(prefix):   # kernel names carry their template arguments ("k_pwg_first<false>")
        ks = [k for k in d if k == prefix or k.startswith(prefix + "<")]
        return d[sorted(ks)[0]] if ks else None
    L, F = d[LK], pick("k_pwg_first")
    Z = pick("k_pwg_last_h3") or pick("k_pwg_last")
    n = 32 * 163840
    if F is not None:
        wcal = F["WRITE_SIZE"] * 1024 / (64 * 4 * n)
    else:
        # round 6: the noise-fed first block -- no k_pwg_first; its own launch writes exactly the x_out planes and the skip, 2 x 64 x 4 B per sample
        L0 = [k for k in d if k.startswith("k_pwg_layer_b3<true")][0]
        wcal = d[L0]["WRITE_SIZE"] * 1024 / (2 * 64 * 4 * n)
    rcal = Z["FETCH_SIZE"] * 1024 / (64 * 4 * n)
    targs = [t.strip() for t in LK[LK.index("<") + 1:LK.rindex(">")].split(",")]   # <FIRST, HALF[, ABL]>
    prof_key = ("pwg_layer_h3" if targs[1] == "true" else "pwg_layer_b3") if "b3" in LK else "pwg_layer"
    extra = {"prof_key": prof_key, "samples_per_launch": n,
             "calibration_note": "FETCH_SIZE calibrated on k_pwg_last[_h3] (reads exactly 64x4 B/sample with the same dword-per-lane, "
                                 "128-B-segment pattern), WRITE_SIZE on k_pwg_first (writes exactly 64x4 B/sample) or, without it, on the noise-fed first block (2x64x4 B/sample); "
                                 "MI355X_MICROARCH.md HBM section: FETCH_SIZE under-counts wide streams by 2x on gfx950"}
else:
    LK = [k for k in d if k.startswith("k_wf_layer_p<2, 3, 0")][0]
    L, P = d[LK], d["k_wf_cond_planes"]
    # pmc_run.py wf 8: 8 utterances of 640 frames -> 10223 folded positions each, framed by gaps of 128, rounded up to a
    # multiple of 128 (pk_wf_infer): 82944 positions = 2592 blocks per row, 16 rows; the conversion kernel reads and rewrites
    # exactly 12288 B per block
    npos = (128 + 8 * (10223 + 128) + 127) // 128 * 128
    known = 16 * (npos // 32) * 12288.0
    wcal = P["WRITE_SIZE"] * 1024 / known
    rcal_planes = P["FETCH_SIZE"] * 1024 / known          # dword-per-lane reads of that kernel (for the record)
    rcal = 0.5                                            # 16 B per lane: the guide's factor
    extra = {"positions_per_launch": npos, "fetch_calibration_dword_kernel": rcal_planes,
             "calibration_note": "FETCH_SIZE x 2 (MI355X_MICROARCH.md: wide 16 B/lane reads are reported at half their bytes on gfx950; "
                                 "the layer kernel reads with 16 B/lane only); WRITE_SIZE calibrated on k_wf_cond_planes (rewrites "
                                 "exactly 12288 B per block)"}
hbm = L["FETCH_SIZE"] * 1024 / rcal + L["WRITE_SIZE"] * 1024 / wcal
out = {"kernel": LK, "hbm_bytes_per_launch": hbm, "fetch_size_kb": L["FETCH_SIZE"], "write_size_kb": L["WRITE_SIZE"],
       "fetch_calibration": rcal, "write_calibration": wcal,
       "l2_hit_rate": L["TCC_HIT"] / (L["TCC_HIT"] + L["TCC_MISS"]), "avg_ns_under_pmc": L["_avg_ns_under_pmc"]}
if "SQ_VALU_MFMA_BUSY_CYCLES" in L:   # (the SQ pass is optional: the traffic passes alone give the bytes)
    out.update({"mfma_busy_frac": L["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * L["GRBM_GUI_ACTIVE"] / 8),
                "effective_clock_ghz": L["GRBM_GUI_ACTIVE"] / 8 / (L["_avg_ns_under_pmc"] * 1e-9) / 1e9,
                "wait_any_frac": L["SQ_WAIT_ANY"] / L["SQ_WAVE_CYCLES"],
                "wait_inst_any_frac": L["SQ_WAIT_INST_ANY"] / L["SQ_WAVE_CYCLES"],
                "valu_quadcycles_per_mfma": L["SQ_ACTIVE_INST_VALU"] / L["SQ_INSTS_MFMA"]})
out.update(extra)
# which kernel source the counters were collected on (bench.py says whether that is the library it timed) and when
import datetime, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parakeet_amd import build as _b
out["kernel_source_sha256"] = _b.file_hash("pwg.hip" if kind == "pwg" else "wf_layer.hip")
out["collected"] = os.environ.get("PK_ROUND", "round 6") + " (" + datetime.date.today().isoformat() + ", tools/pmc_traffic.py)"
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
