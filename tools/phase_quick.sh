#!/bin/bash
# Build box: bash tools/phase_quick.sh build   -> profiling library + profiling specialisation of the puppet
# GPU box:   bash tools/phase_quick.sh         -> per-phase s_memtime cycles of the rollout kernel
F="-DSPEC_ARGS_IN_MEMORY -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form -DTG_PROFILE $TG_EXTRA"
if [ "$1" = build ]; then
    make -s -C trep_amd/csrc prof
    TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F" python -c "
import sys; sys.path.insert(0, '.')
from trep_amd import specialize, systems
print(specialize.build(systems.puppet()))"
else
    TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F" timeout 200 python tools/phase_profile.py "$@"
fi
