#!/bin/bash
# Build every micro-benchmark of this directory for gfx950 into tools/micro/bin/ (git-ignored; travels to the GPU box with gpurun).
set -e
cd "$(dirname "$0")"
mkdir -p bin
for f in *.hip; do
    # gj_bench: sizes as compile-time constants and the flags of the system-specialised kernels (trep_amd/specialize.py)
    X=""
    [ $f = gj_bench.hip ] && X="-DGJ_STATIC_N -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form"
    [ $f = bbd_bench.hip ] && X="-mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I ../../trep_amd/csrc $X $EXTRA -o bin/${f%.hip} $f
done
ls bin
