#!/usr/bin/env python3
"""Cycle shares inside the z-contracted second-derivative kernel (diagnostic build, `make -C trep_amd/csrc prof`):
TREPAMD_LIB=trep_amd/libtrepamd_prof.so python tools/phase_profile_deriv2.py [batch]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd
from trep_amd import systems, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
system = systems.puppet()
Q0 = np.tile(systems.puppet_initial_conditions(system, 64, seed=3), (B // 64 + 1, 1))[:B]
K = systems.puppet_string_schedule(system, Q0[:, system.nQd:], 1, 0.01)
mvi = trep_amd.BatchMidpointVI(system, B)
mvi.initialize_from_configs(0.0, Q0, 0.01, Q0)
mvi.step(0.02, None, K[:, 0])
Z = np.random.default_rng(0).standard_normal((B, mvi.nX))
mvi.timing()
t0 = time.perf_counter(); mvi.deriv2_contract(Z); el = time.perf_counter() - t0
n, ms = mvi.timing()
L = _lib.lib()
out = (ctypes.c_int64 * 16)()
L.tg_batch_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_lib.check(L.tg_batch_profile(mvi._h, out))
v = np.array(list(out)[:12], dtype=float)
print("B=%d  kernel %.2f ms (%.0f /s); cycles of trajectory 0 after the deriv1 solve: %.3e" % (B, ms, B / ms * 1e3, v.sum()))
NAMES = ["constraints: H22 pairs (DDh at q2)", "midpoint evaluation", "third-order body triples", "HZ assembly + store",
         "constraints: seed, w = Kinv' r, sweep q1 + attach", "constraints: prefix / suffix sums along the paths", "constraints: H11 pairs (w-contracted DDDh)",
         "constraints: G1", "constraints: sweep q2 + attach", "HZ: staging TB = H22 Y2 per 16-column block", "HZ: tile chains (matrix cores)", "HZ: epilogue + stores"]
for n_, c in zip(NAMES, v):
    print("  %-46s %12.0f  %5.1f%%" % (n_, c, 100 * c / v.sum()))
