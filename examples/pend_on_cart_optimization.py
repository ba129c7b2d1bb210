#!/usr/bin/env python3
"""examples/pend-on-cart-optimization.py of the reference (BASELINE config 2's system under discopt): swing the pendulum
on a cart along a desired angle profile with DOptimizer (projection-operator descent, quasi-Newton then Newton steps).

    python examples/pend_on_cart_optimization.py
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd as trep
from trep_amd import discopt, systems

t = np.arange(0.0, 10.0, 0.01)
system = systems.pend_on_cart(torque_force=True)      # pend-on-cart-optimization.py:48-64
mvi = trep.MidpointVI(system)
dsys = discopt.DSystem(mvi, t)

# initial trajectory: the uncontrolled system from rest (:66-79)
(X, U) = dsys.build_trajectory()
for k in range(dsys.kf()):
    if k == 0:
        dsys.set(X[k], U[k], 0)
    else:
        dsys.step(U[k])
    X[k + 1] = dsys.f()

# desired trajectory: a smooth 130 degree swing between t = 3 and t = 7 (:81-93)
amp = 130 * math.pi / 180
theta = system.get_config('theta').index
qd = np.zeros((len(t), system.nQ))
inside = (t >= 3.0) & (t <= 7.0)
qd[inside, theta] = (1 - np.cos(2 * math.pi / 4 * (t[inside] - 3.0))) * amp / 2
(Xd, Ud) = dsys.build_trajectory(qd)

Qcost = np.diag([0.01 if i != theta else 100.0 for i in range(dsys.nX)])
Rcost = np.diag([0.01] * dsys.nU)
cost = discopt.DCost(Xd, Ud, Qcost, Rcost)
optimizer = discopt.DOptimizer(dsys, cost, monitor=discopt.DOptimizerVerboseMonitor())
optimizer.first_method_iterations = 4
finished, X, U = optimizer.optimize(X, U, max_steps=12)
print("converged: %s, final cost %.6f, peak angle %.3f rad (desired %.3f)" %
      (finished, optimizer.calc_cost(X, U), np.abs(X[:, theta]).max(), amp))
