#!/bin/bash
# FastSpeech2 feed-forward on planes (ffn_planes.hip) on the GPU box: tests, error vs the fp64 oracle of both paths, A/B timings.
set -u
TAG=${1:-r03ffn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_fs2_gpu.py tests/test_golden_gpu.py tests/test_benchshape_gpu.py -m gpu -q -rA --timeout=300 -k "fs2 or fastspeech or golden" > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -12
timeout 300 python tools/ffn_planes_error.py 37 5 64 1 23 > $OUT/error.log 2>&1; tail -2 $OUT/error.log
run() { local tag=$1; shift; env "$@" timeout 200 python tools/quick_fs2.py > $OUT/quick_$tag.log 2>&1; echo "$tag: $(grep 'FS2 B' $OUT/quick_$tag.log)"; grep -E "ffn|layernorm|bounds" $OUT/quick_$tag.log; }
run planes PK_FS2_FFN_PLANES=1
run gemm PK_FS2_FFN_PLANES=0
run planes_a8 PK_FS2_FFN_PLANES=1 PK_FFNP_ACTIVE=8
run planes_a4 PK_FS2_FFN_PLANES=1 PK_FFNP_ACTIVE=4
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o fs2 -- python $R/tools/quick_fs2.py > $OUT/prof.log 2>&1
python - <<EOF
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    import collections
    d = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if "ffn" in n or "gemm_h3" in n: d[n[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, v in d.items():
        v = v[-64:]
        print(n, len(v), "min %.1f max %.1f us" % (min(v), max(v)), " ".join("%.0f" % x for x in v[-16:]))
EOF
rm -rf $OUT/prof
