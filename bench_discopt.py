#!/usr/bin/env python3
"""bench_discopt.py -- secondary metric of BASELINE.json: discopt iterations/s on the ~40-DOF puppet.

Problem of examples/puppet-optimization.py (reference lines 20-24, 27-105, 127-185): the desired
trajectory is the puppet simulated with its four limb strings moving sinusoidally; the initial guess is
the same puppet with the strings held still; cost weights QD=100, QK=1, PD=1, VK=1, RHO=0.1.  One
"iteration" is one DOptimizer.step: k-parallel linearisation (N DEL solves + deriv1 in one batch),
TV-LQR, [Newton: adjoint + N z-contracted deriv2 in one batch], TV-LQ, forward tangent rollout,
m-parallel Armijo (30 closed-loop N-step rollouts in one batch).

Seeds (perturbed initial poses) are processed one after the other in this round; batching the seeds as
well is the next step.  Prints one JSON line.

  python bench_discopt.py --seeds 4 --horizon 200 --quasi 2 --newton 2
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REFERENCE_NEWTON_ITER_S = 35.0   # reference, puppet, N~1000: >= 35 s per Newton iteration


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--horizon", type=int, default=200, help="number of DEL steps N in the trajectory")
    ap.add_argument("--quasi", type=int, default=2)
    ap.add_argument("--newton", type=int, default=2)
    args = ap.parse_args()

    import trep_amd
    from trep_amd import systems, discopt

    dt, N = 0.01, args.horizon
    system = systems.puppet()
    nd = system.nQd
    t = dt * np.arange(N + 1)
    Q0 = systems.puppet_initial_conditions(system, args.seeds, seed=20250 + 4)
    K_move = systems.puppet_string_schedule(system, Q0[:, nd:], N, dt)
    K_still = np.repeat(Q0[:, None, nd:], N, axis=1)

    sim = trep_amd.BatchMidpointVI(system, args.seeds)
    sim.initialize_from_state(0.0, Q0, np.zeros((args.seeds, nd)))
    Xd = sim.rollout(N, dt, None, K_move)
    sim.initialize_from_state(0.0, Q0, np.zeros((args.seeds, nd)))
    Xi = sim.rollout(N, dt, None, K_still)
    sim.close()

    dsys = discopt.DSystem(trep_amd.MidpointVI(system), t)
    wq = [100.0] * nd + [1.0] * system.nQk + [1.0] * nd + [1.0] * system.nQk
    Qc, Rc = np.diag(wq), np.diag([0.1] * system.nQk)

    results, iters = [], 0
    methods = ['quasi'] * args.quasi + ['newton'] * args.newton
    # warm-up (library load, engine allocation) on seed 0, not timed
    cost = discopt.DCost(Xd[0], K_move[0], Qc, Rc)
    opt = discopt.DOptimizer(dsys, cost)
    opt.step(0, Xi[0].copy(), K_still[0].copy(), 'quasi')
    per_method = {'quasi': [], 'newton': []}
    t0 = time.perf_counter()
    for s in range(args.seeds):
        opt.cost = discopt.DCost(Xd[s], K_move[s], Qc, Rc)
        X, U = Xi[s].copy(), K_still[s].copy()
        c0 = opt.calc_cost(X, U)
        for i, method in enumerate(methods):
            ts = time.perf_counter()
            (done, X, U, dcost0, cost1) = opt.step(i, X, U, method)
            per_method[method].append(time.perf_counter() - ts)
            iters += 1
            if done:
                break
        results.append((c0, opt.calc_cost(X, U)))
    elapsed = time.perf_counter() - t0
    out = {
        "metric": "discopt iterations/s (puppet ~40-DOF, fp64)", "value": iters / elapsed, "unit": "iters/s",
        "n_gpus": 1, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "puppet-optimization.py problem, nX=80 nU=18, N=%d, seeds=%d sequential, %d quasi + %d newton steps each"
                               % (N, args.seeds, args.quasi, args.newton)},
        "mean_s_per_quasi_step": float(np.mean(per_method['quasi'])) if per_method['quasi'] else None,
        "mean_s_per_newton_step": float(np.mean(per_method['newton'])) if per_method['newton'] else None,
        "cost_reduction": [[float(a), float(b)] for a, b in results],
        "reference_s_per_newton_step_N1000": REFERENCE_NEWTON_ITER_S,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
