#!/bin/bash
# Round 6: the noise-fed first block (k_pwg_layer_b3 NZ) -- PWG tests, then batch time with the option on (default) / off, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06n}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x -k "pwg or wavegan or e2e or synth or benchshape or golden or fullsize" --timeout=600 2>&1 | tail -6 | tee $OUT/tests.txt
for rep in 1 2 3; do
  PK_QPWG_NZ=1 timeout 200 python tools/pwg_ab.py noise_fed
  PK_QPWG_NZ=0 timeout 200 python tools/pwg_ab.py first_conv
done 2>&1 | grep -v amdgpu.ids | tee $OUT/pwg_ab.txt
