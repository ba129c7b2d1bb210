#!/usr/bin/env python3
"""GPU box: run BASELINE config 4 (256 puppet seeds x N = 1000), two quasi-Newton and two Newton steps, and save the
iterate, desired trajectory and step record of up to two seeds whose Armijo search is exhausted
(gpurun_out/armijo_failures.npz).  tools/check_armijo_reference.py then replays exactly those steps with the
REFERENCE's DOptimizer in the build container, to see whether the reference fails on them too."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_discopt  # noqa: E402
import trep_amd  # noqa: E402
from trep_amd import discopt  # noqa: E402

S, N, dt = 256, 1000, 0.01
system, Xd, Ud, Xi, Ui, Qc, Rc = bench_discopt.problem(S, N, dt)
dsys = discopt.DSystem(trep_amd.MidpointVI(system), dt * np.arange(N + 1))
opt = discopt.BatchDOptimizer(dsys, Xd, Ud, Qc, Rc)
opt.set_trajectories(Xi, Ui)
out = {}
found = 0
for it, m in enumerate(["quasi", "quasi", "newton", "newton"]):
    X0, U0 = opt.get_trajectories()
    r = opt.step(m)
    bad = np.nonzero(r.failed)[0]
    print("step %d (%s): failed seeds %s; armijo exponents min/median/max %d/%d/%d; fallbacks %d" %
          (it, m, bad.tolist(), r.armijo[~r.failed].min(), np.median(r.armijo[~r.failed]), r.armijo.max(),
           sum(1 for x in r.method if x != m)))
    for s in bad[:2]:
        if found >= 2:
            break
        out["f%d_seed" % found] = np.array([s]); out["f%d_iteration" % found] = np.array([it]); out["f%d_method" % found] = np.array([m])
        out["f%d_X" % found] = X0[s]; out["f%d_U" % found] = U0[s]; out["f%d_Xd" % found] = Xd[s]; out["f%d_Ud" % found] = Ud[s]
        out["f%d_cost0" % found] = np.array([r.cost0[s]]); out["f%d_dcost0" % found] = np.array([r.dcost0[s]])
        out["f%d_final_method" % found] = np.array([r.method[s]])
        found += 1
    # one healthy seed of a Newton step as a control: the reference must accept the same exponent
    if m == "newton" and "c_seed" not in out:
        s = int(np.nonzero(~r.failed)[0][0])
        out["c_seed"] = np.array([s]); out["c_iteration"] = np.array([it]); out["c_method"] = np.array([m])
        out["c_X"] = X0[s]; out["c_U"] = U0[s]; out["c_Xd"] = Xd[s]; out["c_Ud"] = Ud[s]
        out["c_cost0"] = np.array([r.cost0[s]]); out["c_dcost0"] = np.array([r.dcost0[s]]); out["c_cost1"] = np.array([r.cost1[s]])
        out["c_armijo"] = np.array([r.armijo[s]]); out["c_final_method"] = np.array([r.method[s]])
out["Q"] = Qc; out["R"] = Rc; out["t"] = dt * np.arange(N + 1)
np.savez_compressed("gpurun_out/armijo_failures.npz", **out)
print("saved", sorted(out))
opt.close()
