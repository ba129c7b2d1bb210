#!/bin/bash
# gpurun with everything the GPU box needs built first (library, profiling library if present, the test systems' specialisations):
#   bash tools/g.sh [--timeout S] -- 'command'
cd "$(dirname "$0")/.."
make -s -C trep_amd/csrc -j3 2>&1 | grep -E " error|Error [0-9]" && exit 1
python tools/prebuild_specs.py 7 | tail -1
exec gpurun "$@"
