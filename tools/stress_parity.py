#!/usr/bin/env python3
"""One-off randomized parity sweep (GPU box): every test system, several seeds, HIP rollouts / first derivatives /
continuous dynamics against the oracle on random subsets.  Not part of the test-suite; prints one line per system.

    python tools/stress_parity.py [--seeds 5] [--batch 256] [--steps 100]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    import trep_amd
    from common import BUILDERS, build, golden, trajectories, relerr
    from oracle.oracle import OracleMVI, OracleError
    DT = 0.01
    worst_all = 0.0
    for name in sorted(BUILDERS):
        system, d = build(name)
        g = golden(name)
        prefix, q0, U0, K0 = trajectories(name)[0]
        Qg = g[prefix + "Q"]
        nq, nd, nk, nu = d.n_configs, d.n_dyn, d.n_kin, d.n_inputs
        B, N = args.batch, args.steps
        worst = {"rollout": 0.0, "deriv1": 0.0, "dynamics": 0.0}
        fails = 0
        mvi = trep_amd.BatchMidpointVI(system, B)
        o = OracleMVI(d)
        for seed in range(args.seeds):
            rng = np.random.default_rng(1000 + seed)
            # consistent states: points of the recorded reference rollout, restarted at rest (p from q1 = q2)
            idx = rng.integers(0, len(Qg), B)
            Q0 = Qg[idx]
            U = rng.standard_normal((B, N, nu))
            K = Q0[:, None, nd:] + 0.1 * np.sin(2.0 * DT * np.arange(1, N + 1))[None, :, None] * rng.uniform(-1, 1, (B, 1, nk))
            mvi.initialize_from_configs(0.0, Q0, DT, Q0)
            X = mvi.rollout(N, DT, U, K)
            iters, status = mvi.status()
            fails += int((status != 0).sum())
            for b in rng.choice(B, 6, replace=False):
                try:
                    o.initialize_from_configs(0.0, Q0[b], DT, Q0[b])
                    Xo, tot = o.rollout(N, DT, U[b], K[b])
                except OracleError:
                    assert status[b] != 0, (name, seed, b, "oracle failed, HIP did not")
                    continue
                if status[b] == 0:
                    worst["rollout"] = max(worst["rollout"], relerr(X[b], Xo))
            mvi.calc_deriv1()
            b = int(rng.integers(0, B))
            if status[b] == 0:
                o.initialize_from_state((N - 1) * DT, X[b, N - 1, :nq], X[b, N - 1, nq:nq + nd], mvi.lambda1[b] * 0)
                # teacher-forced last step for the derivative comparison
                o.initialize_from_configs(0.0, Q0[b], DT, Q0[b])
                o.rollout(N, DT, U[b], K[b])
                o.calc_deriv1()
                for n in ("q2_dq1", "p2_dq1", "q2_dp1", "p2_dk2", "q2_du1", "l1_dq1"):
                    worst["deriv1"] = max(worst["deriv1"], relerr(mvi.deriv1(n)[b], o.deriv1(n)))
            dQ = rng.standard_normal((B, nq))
            ddK = rng.standard_normal((B, nk))
            ddq, lam, st = mvi.dynamics(Q0, dQ, U[:, 0], ddK)
            for b in rng.choice(B, 6, replace=False):
                f_o, lam_o = o.dynamics(Q0[b], dQ[b], U[b, 0], ddK[b])
                worst["dynamics"] = max(worst["dynamics"], relerr(ddq[b], f_o), relerr(lam[b], lam_o))
        mvi.close()
        worst_all = max(worst_all, *worst.values())
        print("%-16s rollout %.2e  deriv1 %.2e  dynamics %.2e  failed trajectories %d / %d" %
              (name, worst["rollout"], worst["deriv1"], worst["dynamics"], fails, args.seeds * B))
    print("worst relative deviation: %.3e" % worst_all)


if __name__ == "__main__":
    main()
