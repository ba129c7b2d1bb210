#!/bin/bash
# Build-box half of the round-5 collection: everything tools/collect_r05.sh runs on the GPU box must travel with the snapshot
# (the GPU box has hipcc too, but a rocprofv3-wrapped python must not start child processes).
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py > /dev/null
python tools/prebuild_specs.py 7 > /dev/null
make -s -C trep_amd/csrc prof
tools/micro/build.sh > /dev/null
F="-DSPEC_ARGS_IN_MEMORY -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form -DTG_PROFILE"
for extra in "" " -DTG_NO_WEV" " -DTG_NO_WEV -DTG_NO_CMP" " -DSPEC_DERIVATIVES"; do     # (the last one: tools/gpu_prof_derivs.sh)
    TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F$extra" python -c "
import sys; sys.path.insert(0, '.')
from trep_amd import specialize, systems
print(specialize.build(systems.puppet()))"
done
