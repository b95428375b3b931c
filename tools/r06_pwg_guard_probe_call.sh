cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06g3
cp parakeet_amd/libpk_synth_prof.so /tmp/prof_keep.so
{ for rep in 1 2; do timeout 300 python tools/pwg_guard_cost.py product; cp parakeet_amd/variants/amax_probe.so parakeet_amd/libpk_synth_prof.so; PK_PROFILE_LIB=1 timeout 300 python tools/pwg_guard_cost.py amax_probe; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06g3/guard_cost.txt
cp /tmp/prof_keep.so parakeet_amd/libpk_synth_prof.so
