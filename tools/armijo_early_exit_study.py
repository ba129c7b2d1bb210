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
    lamall = opt.armijo_beta ** np.arange(opt.armijo_max_iterations)
    frac = {}; accepted = [None] * S
    searching = [s for s in range(S) if dcost0[s] < 0]
    m0 = 0
    while searching and m0 < opt.armijo_max_iterations:       # chunk after chunk, like step(): every rejected candidate below the accepted one
        costs, ok = opt.armijo_chunk(m0)
        M = costs.shape[1]
        cX = opt.cX.get().reshape(S, M, N + 1, -1); cU = opt.cU.get().reshape(S, M, N, -1)
        for s in list(searching):
            for j in range(M):
                m = m0 + j
                if m >= opt.armijo_max_iterations:
                    break
                bound = cost0[s] + opt.armijo_alpha * lamall[m] * dcost0[s]
                ex = cX[s, j] - Xd[s]; eu = cU[s, j] - Ud[s]
                run = np.cumsum(0.5 * (ex[:-1] ** 2 * qd).sum(1) + 0.5 * (eu ** 2 * rd).sum(1))
                total = run[-1] + 0.5 * (ex[-1] ** 2 * qd).sum()
                if ok[s, j] and total < bound:
                    accepted[s] = m
                    searching.remove(s)
                    break
                k = int(np.argmax(run > bound)) if (run > bound).any() else N     # first step whose running cost is past the bound
                if not ok[s, j]:
                    k = min(k, int(np.argmax(~np.isfinite(run))) if (~np.isfinite(run)).any() else N)
                frac.setdefault(m, []).append((k + 1) / N)
        m0 += M
    allf = np.array([f for v in frac.values() for f in v])
    print("%s step: %d rejected candidates before the accepted one; they pass their bound after %.1f %% of the horizon on average (median %.1f %%, max %.1f %%); accepted exponents %s"
          % (method, len(allf), 100 * allf.mean() if len(allf) else 0, 100 * np.median(allf) if len(allf) else 0, 100 * allf.max() if len(allf) else 0, sorted(set(a for a in accepted if a is not None))))
    print("   by exponent m: mean %% of the horizon before rejection:", {m: round(100 * float(np.mean(v)), 1) for m, v in sorted(frac.items())})
    work = sum(sum(v) for v in frac.values()) + sum(1 for a in accepted if a is not None)
    full = sum(len(v) for v in frac.values()) + sum(1 for a in accepted if a is not None)
    print("   rollout-equivalents with early rejection %.1f vs %d without (%.2fx)" % (work, full, full / max(work, 1e-9)))
    r = opt.step(method)
