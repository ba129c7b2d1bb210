#!/bin/bash
# build-box half: profiling specialisation of the puppet with the current sources (+ optional extra flags), then the phase profile on a GPU box
#   bash tools/prof_r05.sh ["extra flags"]
cd "$(dirname "$0")/.."
make -s -C trep_amd/csrc prof 2>&1 | grep -E "error" 
F="-DSPEC_ARGS_IN_MEMORY -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form -DTG_PROFILE $1"
TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F" python -c "
import sys; sys.path.insert(0, '.')
from trep_amd import specialize, systems
print(specialize.build(systems.puppet()))" || exit 1
gpurun --timeout 600 -- "TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS='$F' timeout 300 python tools/phase_profile.py ${2:-8192} ${3:-50}" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)"
