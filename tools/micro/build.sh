#!/bin/bash
# Build every micro-benchmark of this directory for gfx950 into tools/micro/bin/ (git-ignored; travels to the GPU box with gpurun).
set -e
cd "$(dirname "$0")"
mkdir -p bin
for f in *.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I ../../trep_amd/csrc $EXTRA -o bin/${f%.hip} $f
done
ls bin
