#!/usr/bin/env python3
"""examples/scissor.py of the reference (BASELINE config 5): a four-segment scissor lift -- nine dynamic configs, eight
PointToPoint constraints closing the linkage -- released from a consistent pose and integrated for 10 s; then 4096
lifts with different opening angles as one device-resident rollout.

    python examples/scissor.py
"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd as trep
from trep_amd import systems

dt, tf = 0.01, 10.0
system = systems.scissor_lift(segments=4)     # frames, masses and constraints of scissor.py:16-105
system.satisfy_constraints()                  # scissor.py:104: make the starting guess consistent (SLSQP on the host)
q0 = system.q

mvi = trep.MidpointVI(system)
mvi.initialize_from_configs(0.0, q0, dt, q0)
t0 = time.perf_counter()
q = [mvi.q2]
while mvi.t1 < tf:
    mvi.step(mvi.t2 + dt)
    q.append(mvi.q2)
worst = max(abs(c.h()) for c in system.constraints)
print("scissor lift: %d steps in %.2f s, constraint residual %.1e, L00 %.4f -> %.4f" %
      (len(q) - 1, time.perf_counter() - t0, worst, q[0][0], q[-1][0]))

B, N = 4096, 200
theta = np.random.default_rng(5).uniform(0.03 * math.pi, 0.12 * math.pi, B)
Q0 = np.array([systems.scissor_q(system, th) for th in theta])     # analytic consistent poses (scissor.py:66-97)
batch = trep.BatchMidpointVI(system, B, specialize=True)
batch.initialize_from_configs(0.0, Q0, dt, Q0)
t0 = time.perf_counter()
X = batch.rollout(N, dt)
el = time.perf_counter() - t0
iters, status = batch.status()
print("%d lifts x %d steps: %.3f s (%.2f M DEL steps/s incl. transfers), all converged: %s" %
      (B, N, el, B * N / el / 1e6, bool((status == 0).all())))
