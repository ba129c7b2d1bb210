#!/bin/bash
# Builds tools/micro/bin/gj_bench_<name> for a list of "name:flags" variants (solver experiments), e.g.
#   tools/micro/gj_variants.sh "base:" "nosteps:-DGJP_SKIP_STEPS" "noupdate:-DGJP_SKIP_UPDATE"
cd "$(dirname "$0")"
mkdir -p bin
COMMON="-DTG_GJ_INLINE -DGJ_STATIC_N -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp"
for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I ../../trep_amd/csrc $COMMON $flags -o bin/gj_bench_$name gj_bench.hip &
done
wait
ls bin
