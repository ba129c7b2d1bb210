#!/bin/bash
# phase profile of the specialised rollout kernel with and without the panel solver (diagnostic -DTG_PROFILE builds)
out=gpurun_out/${1:-prof}; mkdir -p $out
F="-DSPEC_ARGS_IN_MEMORY -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -DTG_PROFILE"
TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F" python tools/phase_profile.py > $out/phase_panel.txt 2>&1
TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F -DTG_NO_GJ_PANEL" python tools/phase_profile.py > $out/phase_rows.txt 2>&1
cat $out/phase_panel.txt $out/phase_rows.txt
