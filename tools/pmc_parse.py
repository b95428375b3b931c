"""Average rocprofv3 counter_collection.csv values per kernel.  usage: pmc_parse.py dir [dir...] [--kernel substr]"""
import csv, sys, collections, glob, os, json
args = [a for a in sys.argv[1:] if not a.startswith("--")]
kern = None
for a in sys.argv[1:]:
    if a.startswith("--kernel="): kern = a.split("=", 1)[1]
out = collections.OrderedDict()
for d in args:
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if kern and kern not in name: continue
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[short].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (k, c), v in sorted(agg.items()):
            out.setdefault(k, {})[c] = sum(v) / len(v)
            out[k]["_launches"] = len(v)
            out[k].setdefault("_avg_ns_under_pmc", sum(dur[k]) / len(dur[k]))
print(json.dumps(out, indent=1))
