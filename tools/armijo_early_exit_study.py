#!/usr/bin/env python3
"""How early does a REJECTED Armijo candidate exceed its acceptance bound?  (Would stopping a candidate's projection as soon
as its running cost -- a sum of non-negative terms -- passes cost0 + alpha lambda dcost save work?)
Run on the GPU box: python tools/armijo_early_exit_study.py [seeds] [horizon]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd
from trep_amd import discopt
import bench_discopt

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
system, Xd, Ud, Xi, Ui, Qc, Rc = bench_discopt.problem(S, N, 0.01)
dsys = discopt.DSystem(trep_amd.MidpointVI(system), 0.01 * np.arange(N + 1))
opt = discopt.BatchDOptimizer(dsys, Xd, Ud, Qc, Rc)
opt.set_trajectories(Xi, Ui)
qd, rd = np.diag(Qc), np.diag(Rc)
for method in ("quasi", "newton"):
    opt.linearize(); opt.projection_gain(); cost0 = opt.gradients_and_cost().copy()
    opt.descent_direction(None, method)
    dcost0 = opt.dcost.get().copy()
    costs, ok = opt.armijo_chunk(0)
    M = costs.shape[1]
    cX = opt.cX.get().reshape(S, M, N + 1, -1); cU = opt.cU.get().reshape(S, M, N, -1)
    lam = opt.armijo_beta ** np.arange(M)
    frac = []; accepted = []
    for s in range(S):
        if not (dcost0[s] < 0):
            continue
        acc = None
        for j in range(M):
            bound = cost0[s] + opt.armijo_alpha * lam[j] * dcost0[s]
            ex = cX[s, j] - Xd[s]; eu = cU[s, j] - Ud[s]
            run = np.cumsum(0.5 * (ex[:-1] ** 2 * qd).sum(1) + 0.5 * (eu ** 2 * rd).sum(1))
            total = run[-1] + 0.5 * (ex[-1] ** 2 * qd).sum()
            assert abs(total - costs[s, j]) < 1e-6 * max(1.0, abs(total)) or not ok[s, j], (total, costs[s, j])
            if ok[s, j] and total < bound:
                acc = j
                break
            k = int(np.argmax(run > bound)) if (run > bound).any() else N     # first step whose running cost is past the bound
            frac.append((k + 1) / N)
        accepted.append(acc)
    frac = np.array(frac)
    print("%s step: %d rejected candidates before the accepted one; they pass their bound after %.1f %% of the horizon on average (median %.1f %%, max %.1f %%); accepted exponents %s"
          % (method, len(frac), 100 * frac.mean() if len(frac) else 0, 100 * np.median(frac) if len(frac) else 0, 100 * frac.max() if len(frac) else 0, sorted(set(a for a in accepted if a is not None))))
    r = opt.step(method)
