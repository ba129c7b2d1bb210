#!/usr/bin/env python3
"""Wall time of the forward-mode kernels on the puppet: all fourteen second-derivative arrays of the continuous dynamics of S states
(S x (2 nq + nu) trajectories in one tg_batch_dynamics_deriv1_forward call, host arrays in and out) and the third / fourth-order Lagrangian
derivatives.  python tools/time_forward.py --states 64"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--states", type=int, default=64)
    args = ap.parse_args()
    import trep_amd
    from trep_amd import systems
    system = systems.puppet()
    nq, nk, nu = system.nQ, system.nQk, system.nu
    nv = 2 * nq + nu
    S = args.states
    B = S * nv
    Q0 = systems.puppet_initial_conditions(system, S, seed=5)
    rng = np.random.default_rng(2)
    dQ0, ddK0 = 0.3 * rng.standard_normal((S, nq)), 0.3 * rng.standard_normal((S, nk))
    seed = np.tile(np.concatenate([np.arange(2 * nq), 2 * nq + nk + np.arange(nu)]).astype(np.int32), S)
    rep = lambda a: np.repeat(a, nv, axis=0)
    eng = trep_amd.BatchMidpointVI(system, B)
    out = {"system": "Puppet(string_constraints=True) nq=40 nd=22 nk=18 nc=6", "states": S, "trajectories": B}
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        d, status = eng.dynamics_deriv1(rep(Q0), rep(dQ0), np.zeros((B, nu)), rep(ddK0), seeds=(seed,))
        best = min(best, time.perf_counter() - t0)
    assert (status == 0).all()
    out["dynamics_deriv2_all_arrays"] = {"wall_ms": best * 1e3, "states_per_s": S / best, "note": "host arrays in and out (%.0f MB of results)" % (sum(v.nbytes for v in d.values()) / 1e6)}
    best = 1e9
    idx = rng.integers(0, nq, size=(B, 2)).astype(np.int32)
    for r in range(3):
        t0 = time.perf_counter()
        eng.lagrangian(rep(Q0), rep(dQ0), seeds=(idx[:, 0].copy(), idx[:, 1].copy()))
        best = min(best, time.perf_counter() - t0)
    out["lagrangian_fourth_order"] = {"wall_ms": best * 1e3, "direction_pairs_per_s": B / best}
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
