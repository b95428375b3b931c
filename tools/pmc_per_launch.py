"""Counters and duration of every launch of one kernel, in launch order (rocprofv3 --pmc ... --kernel-trace csv output).
usage: pmc_per_launch.py <dir> [<dir> ...] --kernel=<substr> [--period=30]     (period: fold launch i onto i % period and average)
Round 6: the PWG layer kernel by dilation -- launch i of a call has dilation 2^(i % 10)."""
import csv, sys, glob, os, collections, json
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
kern = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--kernel=")][0]
period = int(([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--period=")] or ["0"])[0])
rows = collections.OrderedDict()
for d in dirs:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]: continue
            e = per.setdefault(int(r["Dispatch_Id"]), {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for i, (_, e) in enumerate(sorted(per.items())):
            k = i % period if period else i
            dst = rows.setdefault(k, collections.defaultdict(list))
            for c, v in e.items(): dst[c if c != "ns" else "ns_" + os.path.basename(d.rstrip("/"))].append(v)
out = [dict(launch=k, **{c: sum(v) / len(v) for c, v in e.items()}) for k, e in rows.items()]
cols = ["launch"] + sorted({c for o in out for c in o if c != "launch"})
print("  ".join(f"{c:>14s}" for c in cols))
for o in out: print("  ".join(f"{o.get(c, float('nan')):14.1f}" if c != "launch" else f"{o[c]:14d}" for c in cols))
