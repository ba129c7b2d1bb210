#!/usr/bin/env python3
"""The reference's examples/extensor-tendon-model.py through the drop-in API: a planar network of nine linear springs
pulled by three constant muscle forces settles into a steady state; then 8192 tendons released from perturbed poses
as one device-resident rollout.  Exercises LinearSpring, HybridWrench and Damping (no gravity, no constraints).

    python examples/extensor_tendon.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd as trep
from trep_amd import systems

tf, dt = 10.0, 0.01
system = systems.extensor_tendon()        # frames, springs, damping, muscle forces and initial pose of the script
q0 = system.q

mvi = trep.MidpointVI(system)
mvi.initialize_from_configs(0.0, q0, dt, q0)
t0 = time.perf_counter()
q = [mvi.q2]
while mvi.t1 < tf:
    mvi.step(mvi.t2 + dt)
    q.append(mvi.q2)
drift = np.abs(q[-1] - q[-101]).max()
print("single tendon: %d steps in %.2f s; change of q over the last second: %.2e (slowly settling), x-3 = %.4f" %
      (len(q) - 1, time.perf_counter() - t0, drift, q[-1][system.get_config('x-3').index]))

# continuous-time view of the final state: the accelerations are small
system.q, system.dq = q[-1], (q[-1] - q[-2]) / dt
print("max |ddq| at the final state: %.2e" % np.abs(system.f()).max())

B, N = 8192, int(tf / dt)
rng = np.random.default_rng(0)
Q0 = q0[None] + 0.1 * rng.standard_normal((B, system.nQ))
batch = trep.BatchMidpointVI(system, B)
batch.initialize_from_configs(0.0, Q0, dt, Q0)
t0 = time.perf_counter()
X = batch.rollout(N, dt)
iters, status = batch.status()
spread = np.abs(X[:, -1, :system.nQ] - q[-1][None]).max()
print("batch: %d tendons x %d steps in %.2f s (incl. transfers), failed: %d, Newton iterations/step %.2f, "
      "max distance of the final poses from the single run's: %.2e" %
      (B, N, time.perf_counter() - t0, int((status != 0).sum()), iters.mean() / N, spread))
