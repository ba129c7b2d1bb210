#!/usr/bin/env python3
"""Host-side profile of one batched Newton step of the puppet discopt problem (where does the wall time outside the kernels go?).
Run on the GPU box:  python tools/profile_discopt_host.py [seeds] [horizon]"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd
from trep_amd import discopt, _lib
import bench_discopt

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
system, Xd, Ud, Xi, Ui, Qc, Rc = bench_discopt.problem(S, N, 0.01)
dsys = discopt.DSystem(trep_amd.MidpointVI(system), 0.01 * np.arange(N + 1))
opt = discopt.BatchDOptimizer(dsys, Xd, Ud, Qc, Rc)
L = _lib.lib()
opt.set_trajectories(Xi, Ui)
opt.step("quasi")
opt.set_trajectories(Xi, Ui)
L.tg_device_synchronize(0)
for m in ("quasi", "newton"):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    r = opt.step(m)
    L.tg_device_synchronize(0)
    pr.disable()
    print("==== %s step: %.3f s, armijo exponents: min %d max %d, failed %d" % (m, time.perf_counter() - t0, r.armijo.min(), r.armijo.max(), int(r.failed.sum())))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
