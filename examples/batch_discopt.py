#!/usr/bin/env python3
"""Trajectory optimisation of the reference's examples/puppet-optimization.py for several seeds at once with the
device-resident BatchDOptimizer: desired motion = strings moving sinusoidally, initial guess = strings held still.

    python examples/batch_discopt.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd as trep
from trep_amd import systems, discopt

S, N, dt = 8, 300, 0.01
system = systems.puppet()
nd = system.nQd
Q0 = systems.puppet_initial_conditions(system, S, seed=1)
K_move = systems.puppet_string_schedule(system, Q0[:, nd:], N, dt)
K_still = np.repeat(Q0[:, None, nd:], N, axis=1)
sim = trep.BatchMidpointVI(system, S)
sim.initialize_from_state(0.0, Q0, np.zeros((S, nd)))
Xd = sim.rollout(N, dt, None, K_move)      # desired trajectories
sim.initialize_from_state(0.0, Q0, np.zeros((S, nd)))
Xi = sim.rollout(N, dt, None, K_still)     # initial guesses
sim.close()

dsys = discopt.DSystem(trep.MidpointVI(system), dt * np.arange(N + 1))
weights = [100.0] * nd + [1.0] * system.nQk + [1.0] * nd + [1.0] * system.nQk
opt = discopt.BatchDOptimizer(dsys, Xd, K_move, np.diag(weights), np.diag([0.1] * system.nQk))
opt.set_trajectories(Xi, K_still)
active = np.ones(S, dtype=bool)
t0 = time.perf_counter()
for i in range(12):
    r = opt.step(opt.select_method(i), active)
    active &= ~r.done
    print("iteration %2d (%s): mean cost %.6g -> %.6g, %d seeds still active" %
          (i, opt.select_method(i), np.nanmean(r.cost0), np.nanmean(r.cost1), int(active.sum())))
    if not active.any():
        break
print("%.2f s" % (time.perf_counter() - t0))
opt.close()
