#!/usr/bin/env python3
"""One-off randomized parity sweep (GPU box): every test system, several seeds, HIP rollouts / first derivatives /
continuous dynamics against the oracle on random subsets.  Not part of the test-suite; prints one line per system
and kernel variant (generic / specialised kernel, default / exact pivot rule) with the worst deviation of the
configurations q, the momenta p and the multipliers lambda separately (relative to each array's largest entry,
`tests/common.py::relerr`).

    python tools/stress_parity.py [--seeds 3] [--batch 128] [--steps 200] [--variants all|default] [--systems a,b]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VARIANTS = (("spec", "default"), ("spec", "exact"), ("generic", "default"), ("generic", "exact"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--checked", type=int, default=6, help="oracle trajectories per seed")
    ap.add_argument("--variants", default="all")
    ap.add_argument("--systems", default="")
    args = ap.parse_args()
    import trep_amd
    from common import BUILDERS, build, golden, trajectories, relerr
    from oracle.oracle import OracleMVI, OracleError
    DT = 0.01
    variants = VARIANTS if args.variants == "all" else (("spec", "default"),)
    names = [s for s in args.systems.split(",") if s] or sorted(BUILDERS)
    worst_all = {"q": 0.0, "p": 0.0, "lambda": 0.0, "deriv1": 0.0, "dynamics": 0.0}
    print("python tools/stress_parity.py --seeds %d --batch %d --steps %d   (HIP vs oracle; relative to each array's largest entry)"
          % (args.seeds, args.batch, args.steps))
    for name in names:
        system, d = build(name)
        g = golden(name)
        prefix, q0, U0, K0 = trajectories(name)[0]
        Qg = g[prefix + "Q"]
        nq, nd, nk, nu, nc = d.n_configs, d.n_dyn, d.n_kin, d.n_inputs, d.n_constraints
        B, N = args.batch, args.steps
        o = OracleMVI(d)
        # the oracle trajectories are shared by the variants: same seeds, same inputs
        cases = []
        for seed in range(args.seeds):
            rng = np.random.default_rng(1000 + seed)
            idx = rng.integers(0, len(Qg), B)       # consistent states: points of the recorded reference rollout, restarted at rest
            Q0 = Qg[idx]
            U = rng.standard_normal((B, N, nu))
            K = Q0[:, None, nd:] + 0.1 * np.sin(2.0 * DT * np.arange(1, N + 1))[None, :, None] * rng.uniform(-1, 1, (B, 1, nk))
            picks = [int(b) for b in rng.choice(B, min(args.checked, B), replace=False)]
            refs = {}
            for b in picks:
                try:
                    o.initialize_from_configs(0.0, Q0[b], DT, Q0[b])
                    Xo, tot = o.rollout(N, DT, U[b], K[b])
                    refs[b] = (Xo, np.array(o.lambda1, dtype=float).copy())
                except OracleError:
                    refs[b] = None
            bd = picks[0]
            d1 = None
            if refs[bd] is not None:
                o.initialize_from_configs(0.0, Q0[bd], DT, Q0[bd])
                o.rollout(N, DT, U[bd], K[bd])
                o.calc_deriv1()
                d1 = {n: np.array(o.deriv1(n)).copy() for n in ("q2_dq1", "p2_dq1", "q2_dp1", "p2_dk2", "q2_du1", "l1_dq1")}
            dQ = rng.standard_normal((B, nq))
            ddK = rng.standard_normal((B, nk))
            dyn = {b: o.dynamics(Q0[b], dQ[b], U[b, 0], ddK[b]) for b in picks}
            cases.append((Q0, U, K, picks, refs, bd, d1, dQ, ddK, dyn))
        for kernel, pivot in variants:
            mvi = trep_amd.BatchMidpointVI(system, B, specialize=(True if kernel == "spec" else False))
            mvi.exact_pivot = (pivot == "exact")
            worst = {"q": 0.0, "p": 0.0, "lambda": 0.0, "deriv1": 0.0, "dynamics": 0.0}
            fails = 0
            its = 0
            for (Q0, U, K, picks, refs, bd, d1, dQ, ddK, dyn) in cases:
                mvi.initialize_from_configs(0.0, Q0, DT, Q0)
                X = mvi.rollout(N, DT, U, K)
                iters, status = mvi.status()
                its += int(iters.sum())
                fails += int((status != 0).sum())
                lam = np.array(mvi.lambda1).reshape(B, nc)
                for b in picks:
                    if refs[b] is None:
                        assert status[b] != 0, (name, kernel, pivot, b, "oracle failed, HIP did not")
                        continue
                    if status[b] != 0:
                        continue
                    Xo, lo = refs[b]
                    worst["q"] = max(worst["q"], relerr(X[b][:, :nq], Xo[:, :nq]))
                    if nd:
                        worst["p"] = max(worst["p"], relerr(X[b][:, nq:nq + nd], Xo[:, nq:nq + nd]))
                    if nc:
                        worst["lambda"] = max(worst["lambda"], relerr(lam[b], lo))
                if d1 is not None and status[bd] == 0:
                    mvi.calc_deriv1()
                    for n, ref in d1.items():
                        if ref.size:
                            worst["deriv1"] = max(worst["deriv1"], relerr(mvi.deriv1(n)[bd], ref))
                ddq, lam_c, st = mvi.dynamics(Q0, dQ, U[:, 0], ddK)
                for b in picks:
                    f_o, lam_o = dyn[b]
                    worst["dynamics"] = max(worst["dynamics"], relerr(ddq[b], f_o), relerr(lam_c[b], lam_o))
            kinfo = mvi.kernel_info()
            ran = "spec" if "rollout" in kinfo.get("spec_launched", ()) else "generic"
            assert ran == kernel, (name, kernel, kinfo)
            mvi.close()
            for k in worst_all:
                worst_all[k] = max(worst_all[k], worst[k])
            print("%-20s %-7s %-7s q %.2e  p %.2e  lambda %.2e  deriv1 %.2e  dynamics %.2e  its/step %.3f  failed %d / %d" %
                  (name, kernel, pivot, worst["q"], worst["p"], worst["lambda"], worst["deriv1"], worst["dynamics"],
                   its / float(args.seeds * B * N), fails, args.seeds * B))
            sys.stdout.flush()
    print("worst over all systems and variants: " + "  ".join("%s %.3e" % kv for kv in worst_all.items()))


if __name__ == "__main__":
    main()
