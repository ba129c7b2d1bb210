F="-DSPEC_ARGS_IN_MEMORY -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -DTG_PROFILE"
mkdir -p gpurun_out/r04
(echo "# system-specialised rollout kernel, structured Newton solve gj_bbd (profiling build: -DTG_PROFILE adds s_memtime stamps)"; TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F" timeout 200 python tools/phase_profile.py;
 echo; echo "# the same with gj_panel (-DTG_NO_BBD: round 3's solver)"; TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F -DTG_NO_BBD" timeout 200 python tools/phase_profile.py) > gpurun_out/r04/phase_profile.txt 2>&1
cat gpurun_out/r04/phase_profile.txt
