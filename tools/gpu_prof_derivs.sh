#!/bin/bash
# cycle shares inside the specialised derivative kernels (diagnostic -DTG_PROFILE builds made on the build box first)
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-prof_derivs}; mkdir -p $out
export TREPAMD_LIB=trep_amd/libtrepamd_prof.so
export TREPAMD_SPEC_FLAGS="-DSPEC_ARGS_IN_MEMORY -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -DTG_PROFILE -DSPEC_DERIVATIVES"    # = tools/prepare_r03.sh
timeout 600 python tools/phase_profile_deriv2.py 8192 > $out/deriv2.txt 2>&1
timeout 600 python tools/phase_profile_deriv1.py 8192 > $out/deriv1.txt 2>&1
cat $out/deriv2.txt $out/deriv1.txt
