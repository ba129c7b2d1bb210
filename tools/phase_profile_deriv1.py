#!/usr/bin/env python3
"""Cycle shares inside the first-derivative kernel (diagnostic build):
TREPAMD_LIB=trep_amd/libtrepamd_prof.so python tools/phase_profile_deriv1.py [batch]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd
from trep_amd import systems, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
system = systems.puppet()
Q0 = np.tile(systems.puppet_initial_conditions(system, 64, seed=3), (B // 64 + 1, 1))[:B]
K = systems.puppet_string_schedule(system, Q0[:, system.nQd:], 1, 0.01)
mvi = trep_amd.BatchMidpointVI(system, B)
mvi.initialize_from_configs(0.0, Q0, 0.01, Q0)
mvi.step(0.02, None, K[:, 0])
mvi.timing()
mvi.calc_deriv1()
n, ms = mvi.timing()
L = _lib.lib()
out = (ctypes.c_int64 * 16)()
L.tg_batch_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_lib.check(L.tg_batch_profile(mvi._h, out))
v = np.array(list(out), dtype=float)
NAMES = ["setup", "pose sweep (mid)", "attach+jacobians", "velocities", "residual", "pose sweeps (q1,q2)",
         "attach+constraints (+constraint Hessian)", "-", "item pairs -> KKT / tables", "GJ scales", "GJ pivot search + swap", "GJ eliminate",
         "-", "output", "-", "-"]
print("B=%d  kernel %.2f ms (%.0f /s); cycles of trajectory 0: %.3e" % (B, ms, B / ms * 1e3, v.sum()))
for n_, c in zip(NAMES, v):
    if c:
        print("  %-44s %12.0f  %5.1f%%" % (n_, c, 100 * c / v.sum()))
