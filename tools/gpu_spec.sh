#!/bin/bash
# generic vs system-specialised rollout kernel: bit-identity check + bench lines
mkdir -p gpurun_out
python - <<'PY'
import numpy as np, trep_amd
from trep_amd import systems
for name, system, B, N in (("puppet", systems.puppet(), 256, 50), ("scissor", systems.scissor_lift(4), 128, 50), ("cart", systems.pend_on_cart(), 300, 50)):
    nd = system.nQd
    if name == "puppet":
        Q0 = systems.puppet_initial_conditions(system, B, seed=5); K = systems.puppet_string_schedule(system, Q0[:, nd:], N, 0.01); U = None
    elif name == "scissor":
        th = np.random.default_rng(1).uniform(0.03*np.pi, 0.12*np.pi, B); Q0 = np.array([systems.scissor_q(system, t) for t in th]); K = None; U = None
    else:
        rng = np.random.default_rng(2); Q0 = np.stack([rng.uniform(-1,1,B), rng.uniform(-3,3,B)], 1); U = rng.standard_normal((B, N, 1)); K = None
    out = []
    for spec in (False, True):
        m = trep_amd.BatchMidpointVI(system, B)
        if spec: m.specialize()
        m.initialize_from_configs(0.0, Q0, 0.01, Q0)
        X = m.rollout(N, 0.01, U, K); it, st = m.status(); out.append((X, it, st)); m.close()
    print(name, "bit-identical:", np.array_equal(out[0][0], out[1][0]), np.array_equal(out[0][1], out[1][1]), "status ok:", (out[1][2] == 0).all())
PY
for flag in "" "--specialize"; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt $flag > gpurun_out/spec_bench$flag.json 2> gpurun_out/spec_bench$flag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/spec_bench$flag.json"))
    print("$flag value %.4g steps/s  kernel %.2f ms  its/step %.3f  %s" % (d["value"], d["roofline"]["kernel_avg_ms"], d["config"]["newton_iterations_per_step"], d["config"]["kernel_variant"]))
except Exception as e:
    print("$flag no bench json", e); print(open("gpurun_out/spec_bench$flag.err").read()[-1500:])
PY
done
