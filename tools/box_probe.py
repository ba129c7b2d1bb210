#!/usr/bin/env python3
"""What kind of box is this?  The MI355X boxes of the pool differ by up to 40 % on some kernels (box-to-box spread):
the open-loop rollout and the first-derivative kernel, one JSON line (bench_discopt.py --stages shows the rest)."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd
from trep_amd import systems, _lib
from trep_amd.discopt.batch_doptimizer import _DevicePool
L = _lib.lib()
out = {}
system = systems.puppet()
B, N = 2048, 200
Q0 = systems.puppet_initial_conditions(system, B, seed=3)
K = systems.puppet_string_schedule(system, Q0[:, system.nQd:], N, 0.01)
mvi = trep_amd.BatchMidpointVI(system, B)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        mvi.timing()
        fn()
        _, ms = mvi.timing()
        best = min(best, ms)
    return best
def open_loop():
    mvi.initialize_from_configs(0.0, Q0, 0.01, Q0)
    mvi.rollout(N, 0.01, None, K)
out["open_loop_rollout_ms_2048x200"] = timed(open_loop)
B2 = 65536
Q1 = np.tile(Q0[:64], (B2 // 64, 1))
K1 = systems.puppet_string_schedule(system, Q1[:, system.nQd:], 1, 0.01)
m2 = trep_amd.BatchMidpointVI(system, B2)
m2.initialize_from_configs(0.0, Q1, 0.01, Q1)
m2.step(0.02, None, K1[:, 0])
m2.timing(); m2.calc_deriv1(); _, ms = m2.timing(); out["deriv1_ms_65536"] = ms
print(json.dumps(out))
