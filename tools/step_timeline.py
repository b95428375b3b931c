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

# ---- whole steps (start of a stack to the start of the next): GPU idle time and where it sits
print()
for gi in range(len(groups) - 1):
    a0, a1 = groups[gi][0][0], groups[gi + 1][0][0]
    ks = sorted((max(s, a0), min(e, a1), n) for s, e, n in K if e > a0 and s < a1)
    cur, idle, holes = a0, 0, []
    for s, e, n in ks:
        if s > cur:
            idle += s - cur
            if s - cur > 3000: holes.append((round((s - cur) / 1e3), n[:28]))
        cur = max(cur, e)
    between = [(s, e, n) for s, e, n in ks if s >= groups[gi][-1][1]]
    bsum = collections.Counter()
    for s, e, n in between: bsum[n[:28]] += e - s
    print(f"step {1e-6*(a1-a0):7.3f} ms: stack {1e-6*(groups[gi][-1][1]-a0):7.3f}, after it {1e-6*(a1-groups[gi][-1][1]):6.3f} ms of which GPU idle {1e-6*idle:6.3f} ms; holes > 3 us (us, next kernel): {holes[:10]}")
    print("    kernel time after the stack (us, summed, overlaps counted twice):", {k: round(v / 1e3) for k, v in bsum.most_common(10)}, " total", round(sum(bsum.values()) / 1e3))

# ---- neighbourhood of the big holes of one step
gi = max(0, len(groups) - 4)
a0, a1 = groups[gi][-1][1] - 50_000, groups[gi + 1][0][0] + 50_000
ks = sorted((s, e, n) for s, e, n in K if e > a0 and s < a1)
print("\nkernels from 50 us before the end of a stack to the start of the next (start us, duration us, name); '<<' marks a hole > 20 us before the kernel")
cur = ks[0][1]
for s, e, n in ks:
    mark = f"  << {round((s - cur) / 1e3)} us idle" if s - cur > 20_000 else ""
    print(f"  {1e-3*(s-a0):9.1f} {1e-3*(e-s):8.1f}  {n[:60]}{mark}")
    cur = max(cur, e)
