import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, trep_amd
from trep_amd import systems
from common import build
system, d = build("puppet40")
B, N = 16, 10
Q0 = systems.puppet_initial_conditions(system, B, seed=3)
for dt in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
    K = systems.puppet_string_schedule(system, Q0[:, system.nQd:], N, dt)
    mvi = trep_amd.BatchMidpointVI(system, B, specialize=True)
    mvi.initialize_from_configs(0.0, Q0, dt, Q0)
    X = mvi.rollout(N, dt, None, K)
    it, st = mvi.status(); fb = mvi.solver_fallbacks()
    print(dt, "status", st.max(), "iters", it.sum(), "fallbacks", fb.sum())
    mvi.close()
