#!/bin/bash
# Round 5: a non-deterministic 64-channel WaveFlow call under variant libraries (tools/build_variant.py, tools/asm_variant.py).
# usage: VARIANTS="name ..." [WF_FRAMES=..] [WF_VARIANTS=default,waves8,..] [WF_REP=n] tools/r05_wf_race_call.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export WF_VARIANTS=${WF_VARIANTS:-default,waves8}
run() { echo "== $1"; shift; timeout 200 "$@" python tools/wf_race_bisect.py 2>&1 | grep -v amdgpu | tail -n +2; }
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
run product env
for v in ${VARIANTS:-}; do
  cp parakeet_amd/variants/$v.so parakeet_amd/libpk_synth_prof.so
  run "variant $v" env PK_PROFILE_LIB=1
done
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
