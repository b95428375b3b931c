#!/bin/bash
# hop-300 Parallel WaveGAN (baker / vctk upsample_scales): tests, then a one-box A/B of the layer kernel against the previous pwg.hip
# (variant library), hop 300 and hop 256 (the headline kernel must not move).   usage: tools/r05_pwg_gen_call.sh <tag>
set -u
TAG=${1:-r05o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 700 python -m pytest tests/test_pwg_gpu.py tests/test_speedyspeech_gpu.py -m gpu -q --timeout=300 2>&1 | tail -3) > $OUT/tests.txt
tail -2 $OUT/tests.txt
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
run() { PK_QPWG_SCALES=$2 timeout 150 python tools/quick_pwg.py 2>&1 | grep -E "^PWG|pwg_layer" | tr '\n' ' ' | sed "s/^/$1: /"; echo; }
{
for rep in 1 2 3; do
  for sc in "4,5,3,5" ""; do
    run product "$sc"
    cp parakeet_amd/variants/pwg_r04.so parakeet_amd/libpk_synth_prof.so
    PK_PROFILE_LIB=1 run pwg_r04 "$sc"
  done
done
} > $OUT/pwg_ab.txt 2>&1
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
cat $OUT/pwg_ab.txt
