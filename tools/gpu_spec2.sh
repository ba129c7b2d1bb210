#!/bin/bash
# specialised kernel variants: one bench line per flag set (flags are part of the cache key; build them in the container first)
mkdir -p gpurun_out
for fl in "$@"; do
  export TREPAMD_SPEC_FLAGS="$fl"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt > gpurun_out/spec2.json 2> gpurun_out/spec2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/spec2.json"))
    print("[$fl] value %.4g steps/s  kernel %.2f ms  its/step %.3f failed %d %s" % (d["value"], d["roofline"]["kernel_avg_ms"], d["config"]["newton_iterations_per_step"], d["config"]["failed_trajectories"], d["config"]["kernel_variant"]))
except Exception as e:
    print("[$fl] no bench json", e); print(open("gpurun_out/spec2.err").read()[-800:])
PY
done
