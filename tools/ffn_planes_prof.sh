#!/bin/bash
export PK_PROFILE_LIB=1   # the PK_* measurement switches exist in the profile build only: python -m parakeet_amd.build --profile (libpk_synth_prof.so)
# per-launch durations of the feed-forward kernels of tools/quick_fs2.py (encoder and decoder launches apart).
# usage: ffn_planes_prof.sh <tag> <case>...   case = gemm | <variant>[:<ablate>]   (PK_FFNP_VARIANT / PK_FFNP_ABLATE)
set -u
TAG=${1:-r03ffnp}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, 'CUs')"
for c in "$@"; do
v=${c%%:*}; a=0; [[ "$c" == *:* ]] && a=${c##*:}
if [ "$v" = gemm ]; then export PK_FS2_FFN_PLANES=0; else export PK_FS2_FFN_PLANES=1 PK_FFNP_VARIANT=$v PK_FFNP_ABLATE=$a; fi; [ "$v" = 0 ] && unset PK_FFNP_VARIANT
{ [ "$a" = 0 ] || [ "$a" = 1024 ]; } && echo "   error vs fp64 oracle: $(timeout 100 python $R/tools/ffn_planes_error.py 37 5 2>&1 | grep planes)"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$c -o fs2 -- python $R/tools/quick_fs2.py > $OUT/prof_$c.log 2>&1
echo "== $c: $(grep 'FS2 B' $OUT/prof_$c.log)"
python - <<EOF
import csv, glob, collections
f = glob.glob("$OUT/prof_$c/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "ffn" in n or "gemm_h3" in n or "attention" in n: d[n[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    print("  ", n.replace("(anonymous namespace)::", "")[:44], len(v), " ".join("%.0f" % x for x in v[-8:]))
EOF
rm -rf $OUT/prof_$c
done
