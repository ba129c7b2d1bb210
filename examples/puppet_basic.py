#!/usr/bin/env python3
"""The marionette of the reference's examples/puppet-basic.py through the drop-in API: build the system, make
the starting guess consistent with the six string constraints, integrate with the single-trajectory
MidpointVI (B = 1 shell over the batched HIP engine), then run 4096 perturbed copies as one device-resident rollout.

    python examples/puppet_basic.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd as trep
from trep_amd import systems

dt, tf = 0.01, 2.0
system = systems.puppet_basic()           # same frame tree, masses, damping and Distance constraints as the script
system.q = systems.PUPPET_BASIC_POSE      # the script's starting guess ...
system.satisfy_constraints()              # ... made consistent with the strings
q0 = system.q

mvi = trep.MidpointVI(system)
mvi.initialize_from_configs(0.0, q0, dt, q0)
t0 = time.perf_counter()
q = [mvi.q2]
while mvi.t1 < tf:
    mvi.step(mvi.t2 + dt, (), system.qk)
    q.append(mvi.q2)
print("single trajectory: %d steps in %.2f s, TorsoZ %.4f -> %.4f" %
      (len(q) - 1, time.perf_counter() - t0, q[0][system.get_config('TorsoZ').index], q[-1][system.get_config('TorsoZ').index]))

# the same system, 4096 perturbed poses, one kernel launch
B, N = 4096, int(tf / dt)
rng = np.random.default_rng(0)
Q0 = np.repeat(q0[None], B, axis=0)
for b in range(1, 8):                      # a few distinct consistent poses, tiled over the batch
    system.q = q0 + rng.uniform(-0.03, 0.03, system.nQ) * (np.arange(system.nQ) >= 3)
    Q0[b::8] = system.satisfy_constraints()
batch = trep.BatchMidpointVI(system, B)
batch.initialize_from_configs(0.0, Q0, dt, Q0)
t0 = time.perf_counter()
X = batch.rollout(N, dt)                   # [B][N+1][nX]
iters, status = batch.status()
print("batch: %d trajectories x %d steps in %.2f s (incl. transfers), failed: %d, Newton iterations/step %.2f" %
      (B, N, time.perf_counter() - t0, int((status != 0).sum()), iters.mean() / N))
