"""Where the pipelined step's time goes: from a rocprofv3 --kernel-trace csv of bench.py, the gaps between consecutive PWG layer launches and what ran in them.
usage: step_timeline.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]) for r in csv.DictReader(open(f))]
K.sort()
layers = [k for k in K if k[2].startswith("k_pwg_layer_b3")]
# steps: groups of 30 consecutive layer launches; take the last 8 groups
groups = [layers[i:i + 30] for i in range(0, len(layers) - 29, 30)][-8:]
for g in groups:
    t0, t1 = g[0][0], g[-1][1]
    busy = sum(e - s for s, e, _ in g)
    gaps = [(g[i + 1][0] - g[i][1]) for i in range(29)]
    inside = collections.Counter()
    for s, e, n in K:
        if n.startswith("k_pwg_layer_b3") or e <= t0 or s >= t1: continue
        inside[n[:40]] += (min(e, t1) - max(s, t0))
    big = sorted(gaps, reverse=True)[:5]
    print(f"stack {1e-6*(t1-t0):7.3f} ms  layer kernels {1e-6*busy:7.3f} ms  gaps {1e-6*sum(gaps):6.3f} ms (largest us: {[round(x/1e3) for x in big]})  avg layer {1e-3*busy/30:7.1f} us")
    print("    other kernels' wall time inside the stack (us, overlapping or in gaps):", {k: round(v / 1e3) for k, v in inside.most_common(8)})
