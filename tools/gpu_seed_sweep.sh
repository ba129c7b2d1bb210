#!/bin/bash
# discopt seed sweep (BASELINE config 4 per-GPU shards): 32 / 64 / 128 / 256 seeds, N = 1000, with per-stage times
tag=${1:-r05}
mkdir -p gpurun_out/$tag
for s in 32 64 128 256; do timeout 900 python bench_discopt.py --seeds $s --horizon 1000 --quasi 1 --newton 1 --stages > gpurun_out/$tag/discopt_$s.json 2> gpurun_out/$tag/discopt_$s.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/$tag/discopt_$s.json"))
    print("seeds %4d  %.1f it/s  quasi %.4f s  newton %.4f s  armijo failures %s" % ($s, d["iters_per_s"], d["s_per_batched_quasi_step"], d["s_per_batched_newton_step"], d.get("armijo_failures")))
    st = d.get("stage_seconds_one_newton_step") or d.get("stages")
    if st: print("   ", {k: round(v, 4) for k, v in st.items()})
except Exception as e:
    print("seeds $s failed", e); print(open("gpurun_out/$tag/discopt_$s.err").read()[-1500:])
PY
done
