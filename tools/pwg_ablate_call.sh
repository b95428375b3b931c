#!/bin/bash
export PK_PROFILE_LIB=1   # the PK_* measurement switches exist in the profile build only: python -m parakeet_amd.build --profile (libpk_synth_prof.so)
# Parallel WaveGAN layer kernel, profiling ablations (results wrong by construction): per-launch time of tools/quick_pwg.py
#   PK_PWG_ABLATE: 1 = no global loads / stores, 32 = the x taps without their hi / lo split, 96 / 160 = 32 + the taps loaded
#   as two 16-byte vectors per group (hi | lo adjacent / the planes 512 bytes apart)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03pwgabl}
shift
mkdir -p $OUT
cd $R
for a in "$@"; do
  if [ $a = 0 ]; then unset PK_PWG_ABLATE; else export PK_PWG_ABLATE=$a; fi
  timeout 200 python tools/quick_pwg.py > $OUT/quick_$a.log 2>&1
  echo "ablate $a: $(grep 'PWG B' $OUT/quick_$a.log) | $(grep pwg_layer_h3 $OUT/quick_$a.log)"
done
