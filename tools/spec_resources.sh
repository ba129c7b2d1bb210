#!/bin/bash
# Register / scratch / LDS use of every kernel of a specialised library: recompiles the given cached header with
# -Rpass-analysis=kernel-resource-usage.   tools/spec_resources.sh trep_amd/_spec/libtrepamd_spec_<key>.hpp [extra flags]
hdr=$(readlink -f "$1"); shift
cd "$(dirname "$0")/../trep_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -I . "-DTG_SPEC_HEADER=\"$hdr\"" \
  -DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp "$@" \
  -Rpass-analysis=kernel-resource-usage -o /tmp/spec_resources.so spec_kernel.hip 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|SGPRs:|LDS Size" 
