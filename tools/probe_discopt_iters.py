#!/usr/bin/env python3
"""Newton iteration counts and times of the discopt rollouts (linearisation: one hinted step per (seed, k); Armijo: closed-loop projections)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_discopt
from trep_amd import discopt
import trep_amd

S, N, dt = 32, 200, 0.01
system, Xd, Kd, Xi, Ki, Q, R = bench_discopt.problem(S, N, dt)
mvi = trep_amd.MidpointVI(system)
t = np.arange(N + 1) * dt
dsys = discopt.DSystem(mvi, t)
opt = discopt.BatchDOptimizer(dsys, Xd, Kd, Q, R)
opt.set_trajectories(Xi, Ki)
for rep in range(2):
    t0 = time.perf_counter(); broken = opt.linearize(); opt.lin.synchronize(); el = time.perf_counter() - t0
    it, st = opt.lin.status()
    print("linearize: %.2f ms  Newton iterations / step %.3f  max %d  failed %d   kernel info %s" % (1e3 * el, it.mean(), it.max(), (st != 0).sum(), opt.lin.kernel_info()["spec_launched"]))
opt.projection_gain()
opt.gradients_and_cost()
seeds = np.arange(S)
opt.descent_direction(seeds, "quasi")
for rep in range(2):
    t0 = time.perf_counter(); opt.armijo_chunk(0); opt.arm.synchronize(); el = time.perf_counter() - t0
    it, st = opt.arm.status()
    print("armijo chunk: %.2f ms  total Newton iterations %d  failed %d of %d  iterations of the survivors / step %.3f" %
          (1e3 * el, it.sum(), (st != 0).sum(), len(st), it[st == 0].mean() / N if (st == 0).any() else 0))
