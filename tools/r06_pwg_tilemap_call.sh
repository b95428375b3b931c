#!/bin/bash
# Round 6: tile maps of the PWG layer kernel (PK_PWG_TILE_MAP variants) -- batch time interleaved with the product, and FETCH_SIZE / TCC_HIT per launch.
#   VARIANTS="tilemap1 tilemap2" bash tools/r06_pwg_tilemap_call.sh <tag> [reps]
set -u
TAG=${1:-r06t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
{
for rep in $(seq 1 ${2:-3}); do
  timeout 200 python tools/pwg_ab.py product
  for v in ${VARIANTS}; do
    cp parakeet_amd/variants/$v.so parakeet_amd/libpk_synth_prof.so
    PK_PROFILE_LIB=1 timeout 200 python tools/pwg_ab.py $v
  done
done
} 2>&1 | grep -v "amdgpu.ids" | tee $OUT/pwg_ab.txt
cd /tmp
for v in product ${VARIANTS}; do
  if [ $v = product ]; then cp /tmp/prof_keep.so $R/parakeet_amd/libpk_synth_prof.so; else cp $R/parakeet_amd/variants/$v.so $R/parakeet_amd/libpk_synth_prof.so; fi
  PK_PROFILE_LIB=1 timeout 240 rocprofv3 --pmc FETCH_SIZE TCC_HIT --kernel-trace --output-format csv -d $OUT/pmc_$v -o p -- python $R/tools/pmc_run.py pwg 32 > $OUT/pmc_$v.log 2>&1
  echo "== $v"; python $R/tools/pmc_per_launch.py $OUT/pmc_$v --kernel=k_pwg_layer --period=10 | tee $OUT/per_launch_$v.txt
done
cp /tmp/prof_keep.so $R/parakeet_amd/libpk_synth_prof.so
find $OUT -maxdepth 1 -type d -name "pmc_*" | xargs rm -rf
