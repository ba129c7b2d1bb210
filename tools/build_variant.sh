#!/bin/bash
# Build an experimental variant of the HIP library next to the product one:
#   tools/build_variant.sh <name> "<extra -D flags>"   ->  trep_amd/libtrepamd_<name>.so   (select with TREPAMD_LIB=...)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../trep_amd/csrc"
mkdir -p build
make -s build/dopt.o build/comm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c -o build/trepamd_$name.o trepamd.hip 2> build/trepamd_$name.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libtrepamd_$name.so build/trepamd_$name.o build/dopt.o build/comm.o -ldl
echo "built libtrepamd_$name.so"
