#!/bin/bash
export PK_PROFILE_LIB=1   # the PK_* measurement switches exist in the profile build only: python -m parakeet_amd.build --profile (libpk_synth_prof.so)
# Parallel WaveGAN with x as pre-split planes (PK_PWG_PLANES) on the GPU box: the PWG tests with it on, then the per-launch
# time of the layer kernel both ways.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03pwgpl}
mkdir -p $OUT
cd $R
PK_PWG_PLANES=1 timeout 600 python -m pytest tests/test_pwg_gpu.py -m gpu -q -x --timeout=300 2>&1 | tail -n 1
for p in 1 0; do
  echo "planes $p: $(PK_PWG_PLANES=$p timeout 200 python tools/quick_pwg.py 2>&1 | grep -E 'PWG B|pwg_layer_h3' | tr '\n' '|')"
done
