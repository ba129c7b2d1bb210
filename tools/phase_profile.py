#!/usr/bin/env python3
"""Per-phase cycle shares of the rollout kernel (diagnostic build: `make -C trep_amd/csrc prof`).
Run on the GPU box:  TREPAMD_LIB=trep_amd/libtrepamd_prof.so python tools/phase_profile.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd
from trep_amd import systems, _lib

NAMES = ["newton update + rates", "pose sweep: chains (dual) | whole (mid)", "attach+jacobians", "velocities", "residual", "sin/cos (dual sweep) | pose sweep (q1/q2)",
         "attach+constraints", "newton init", "newton pairs", "step set-up, result rows | GJ scales (generic assembly)", "GJ pivot+swap", "GJ eliminate",
         "converged?", "tail", "Gauss-Jordan (registers)", "local transforms (dual sweep)"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
system = systems.puppet()
Q0 = np.tile(systems.puppet_initial_conditions(system, 64, seed=3), (B // 64 + 1, 1))[:B]
K = systems.puppet_string_schedule(system, Q0[:, system.nQd:], N, 0.01)
mvi = trep_amd.BatchMidpointVI(system, B)
mvi.initialize_from_configs(0.0, Q0, 0.01, Q0)
K_dev = mvi.device_array(K)
mvi.rollout_device(N, 0.01, None, K_dev, None)
mvi.synchronize()
L = _lib.lib()
out = (ctypes.c_int64 * 16)()
L.tg_batch_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_lib.check(L.tg_batch_profile(mvi._h, out))
v = np.array(list(out), dtype=float)
it, st = mvi.status()
ticks = v[13]; v[13] = 0.0          # rollout kernels: slot 13 = s_memrealtime ticks (100 MHz) of trajectory 0
print("B=%d N=%d  its/step %.2f  total cycles (traj 0) %.3e  per step %.0f   shader clock %.0f MHz (cycles / 100 MHz wall-clock ticks)" % (B, N, it.mean() / N, v.sum(), v.sum() / N, 100.0 * v.sum() / max(ticks, 1.0)))
for n, c in zip(NAMES, v):
    if c:
        print("  %-44s %12.0f  %5.1f%%  %8.0f /step" % (n, c, 100 * c / v.sum(), c / N))
