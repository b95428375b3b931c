#!/bin/bash
# Parallel WaveGAN with x as pre-split planes (PK_PWG_PLANES=1) on the GPU box: the PWG tests and the full-size / bench-shape
# tests with it on, then per-launch time of the layer kernel both ways.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03pwgpl}
mkdir -p $OUT
cd $R
PK_PWG_PLANES=1 timeout 900 python -m pytest tests/test_pwg_gpu.py tests/test_benchshape_gpu.py tests/test_fullsize_gpu.py tests/test_speedyspeech_gpu.py tests/test_golden_gpu.py -m gpu -q -rA --timeout=300 -k "pwg or e2e or baker or golden" > $OUT/tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -12
for p in 0 1 0 1; do
  echo "planes $p: $(PK_PWG_PLANES=$p timeout 200 python tools/quick_pwg.py 2>&1 | grep -E 'PWG B|pwg_layer_h3|pwg_first' | tr '\n' '|')"
done
