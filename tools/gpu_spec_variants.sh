#!/bin/bash
# Rollout bench (no discopt, no CPU baseline) under several specialised-kernel flag sets.  Build the variants first (CPU box):
#   tools/gpu_spec_variants.sh build "name:flags" ...      then on the GPU box:   tools/gpu_spec_variants.sh run <tag> "name:flags" ...
mode=$1; shift
BASE="-DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp"
if [ "$mode" = build ]; then
    for v in "$@"; do
        flags=${v#*:}; [ "${flags#@}" != "$flags" ] && F="${flags#@}" || F="$BASE $flags"
        TREPAMD_SPEC_FLAGS="$F" python -c "
import sys; sys.path.insert(0, '.')
from trep_amd import specialize, systems
print('${v%%:*}', specialize.build(systems.puppet()))" &
    done
    wait
else
    tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
    for rep in 1 2; do
    for v in "$@"; do
        name=${v%%:*}; flags=${v#*:}; [ "${flags#@}" != "$flags" ] && F="${flags#@}" || F="$BASE $flags"
        TREPAMD_SPEC_FLAGS="$F" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-discopt > $out/bench_$name.json 2> $out/bench_$name.err
        python - <<PY
import json
try:
    d = json.load(open("$out/bench_$name.json"))
    print("%-14s %.4g steps/s kernel %.2f ms its/step %.3f failed %d" % ("$name", d["value"], d["roofline"]["kernel_avg_ms"], d["config"]["newton_iterations_per_step"], d["config"]["failed_trajectories"]))
except Exception as e:
    print("$name", "failed", e); print(open("$out/bench_$name.err").read()[-800:])
PY
    done; done
fi
