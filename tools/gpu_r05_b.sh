#!/bin/bash
# full GPU suite on the world-frame evaluation + its phase profile (and the item form beside it, -DTG_NO_WEV)
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05/full_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r05/full_pytest.log | tail -3
F="-DSPEC_ARGS_IN_MEMORY -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form -DTG_PROFILE"
(echo "# system-specialised rollout kernel, world-frame evaluation (profiling build: -DTG_PROFILE adds s_memtime stamps)"
 echo "# stamp names of the world form: 'attach+jacobians' = E3 (body poses, end points, world twists), 'attach+constraints' = E4 (constraints, list sums, body entries), 'velocities' = E5 (group composites), 'residual' = E6, 'newton init' = phase C, 'newton pairs' = phase D"
 TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F" timeout 300 python tools/phase_profile.py
 echo; echo "# the item form of round 4 (-DTG_NO_WEV), same box"
 TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F -DTG_NO_WEV" timeout 300 python tools/phase_profile.py) > gpurun_out/r05/phase_profile.txt 2>&1
cat gpurun_out/r05/phase_profile.txt
