#!/usr/bin/env python3
"""The reference's examples/dual_pendulums.py through the drop-in API: two pendulums joined by a linear spring and a
linear damper; the total energy decays through the damper.  Then 4096 copies released from random angles in one launch.

    python examples/dual_pendulums.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd as trep
from trep_amd import systems

dt, tf = 0.01, 10.0
system = systems.dual_pendulums()          # frames, spring (k = 20, x0 = 1), damper (c = 1), gravity, q = (3, -3)
q0 = system.q

mvi = trep.MidpointVI(system)
mvi.initialize_from_configs(0.0, q0, dt, q0)
t0 = time.perf_counter()
q, t = [mvi.q2], [mvi.t2]
while mvi.t1 < tf:
    mvi.step(mvi.t2 + dt)
    q.append(mvi.q2)
    t.append(mvi.t2)
q = np.array(q)
print("single run: %d steps in %.2f s, final angles %s" % (len(q) - 1, time.perf_counter() - t0, np.round(q[-1], 4)))

# energy along the trajectory (midpoint states), all in one launch of the energy kernel
eng = trep.BatchMidpointVI(system, len(q) - 1)
TV = eng.energy(0.5 * (q[1:] + q[:-1]), (q[1:] - q[:-1]) / dt)
E = TV.sum(axis=1)
print("total energy: %.3f at the start -> %.3f at t = %.0f s (dissipated by the damper)" % (E[0], E[-1], tf))

B, N = 4096, int(tf / dt)
rng = np.random.default_rng(1)
Q0 = rng.uniform(-np.pi, np.pi, (B, 2))
batch = trep.BatchMidpointVI(system, B)
batch.initialize_from_configs(0.0, Q0, dt, Q0)
t0 = time.perf_counter()
X = batch.rollout(N, dt)
iters, status = batch.status()
print("batch: %d pairs of pendulums x %d steps in %.2f s (incl. transfers), failed: %d, Newton iterations/step %.2f" %
      (B, N, time.perf_counter() - t0, int((status != 0).sum()), iters.mean() / N))
