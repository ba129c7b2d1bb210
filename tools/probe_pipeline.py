#!/usr/bin/env python3
"""Times the stages of a Newton step with and without the chunk pipeline (32 seeds by default)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_discopt
from trep_amd import discopt, _lib
import trep_amd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, dt = 1000, 0.01
system, Xd, Kd, Xi, Ki, Q, R = bench_discopt.problem(S, N, dt)
dsys = discopt.DSystem(trep_amd.MidpointVI(system), np.arange(N + 1) * dt)
L = _lib.lib()
for pipe in (False, True):
    opt = discopt.BatchDOptimizer(dsys, Xd, Kd, Q, R, pipeline_newton=pipe)
    opt.set_trajectories(Xi, Ki)
    opt.step("newton")          # warm-up: allocations
    opt.set_trajectories(Xi, Ki)
    def timed(name, fn, *a):
        L.tg_device_synchronize(0); t0 = time.perf_counter(); fn(*a); L.tg_device_synchronize(0)
        print("  pipeline %-5s %-46s %.4f s" % (pipe, name, time.perf_counter() - t0))
    for rep in range(2):
        timed("linearize", opt.linearize)
        timed("cost + gradients", opt.gradients_and_cost)
        if pipe:
            timed("projection | quasi | curvature | newton LQ (pipeline)", opt.projection_quasi_and_newton_model)
        else:
            timed("projection || quasi", opt.projection_gain_and_quasi_direction, True)
            timed("newton direction (curvature, LQ, tangent)", opt.descent_direction, None, "newton")
        timed("armijo chunk", opt.armijo_chunk, 0)
    opt.close()
