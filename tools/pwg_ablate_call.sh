#!/bin/bash
# Parallel WaveGAN layer kernel, profiling ablations (results wrong by construction): per-launch time of tools/quick_pwg.py
#   PK_PWG_ABLATE unset = as built, 1 = no global loads / stores, 32 = the x taps without their hi / lo split
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03pwgabl}
mkdir -p $OUT
cd $R
for a in 0 32 1 0 32; do
  if [ $a = 0 ]; then unset PK_PWG_ABLATE; else export PK_PWG_ABLATE=$a; fi
  timeout 200 python tools/quick_pwg.py > $OUT/quick_$a.log 2>&1
  echo "ablate $a: $(grep 'PWG B' $OUT/quick_$a.log) | $(grep pwg_layer_h3 $OUT/quick_$a.log)"
done
