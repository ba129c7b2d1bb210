#!/bin/bash
# specialised kernel variants (flags are part of the cache key; all were built in the container)
mkdir -p gpurun_out
for fl in "" "-DSPEC_ARGS_IN_MEMORY" "-DTG_NO_DUAL_SWEEP" "-DSPEC_ARGS_IN_MEMORY -DTG_NO_DUAL_SWEEP" "$@"; do
  export TREPAMD_SPEC_FLAGS="$fl"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt --specialize > gpurun_out/spec2.json 2> gpurun_out/spec2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/spec2.json"))
    print("[$fl] value %.4g steps/s  kernel %.2f ms  its/step %.3f failed %d %s" % (d["value"], d["roofline"]["kernel_avg_ms"], d["config"]["newton_iterations_per_step"], d["config"]["failed_trajectories"], d["config"]["kernel_variant"]))
except Exception as e:
    print("[$fl] no bench json", e); print(open("gpurun_out/spec2.err").read()[-1500:])
PY
done
