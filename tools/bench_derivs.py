#!/usr/bin/env python3
"""Throughput of the derivative kernels (SURVEY section 8 rows a-12 .. a-14) on the puppet: teacher-forced DEL
solve, first derivatives written as A/B (tg_batch_linearize), z-contracted second derivatives
(tg_batch_deriv2_contract_device), and of the continuous dynamics (tg_batch_dynamics_device, section 8f rank 2).  Kernel times from the library's HIP events.  Prints one JSON line.

  python tools/bench_derivs.py --batch 65536
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import trep_amd
    from trep_amd import systems, _lib
    from trep_amd.discopt.batch_doptimizer import _DevicePool
    L = _lib.lib()
    system = systems.puppet()
    B, dt, nd = args.batch, 0.01, system.nQd
    distinct = min(B, 256)
    Q0 = np.tile(systems.puppet_initial_conditions(system, distinct, seed=11), ((B + distinct - 1) // distinct, 1))[:B]
    K = systems.puppet_string_schedule(system, Q0[:, nd:], 1, dt)
    mvi = trep_amd.BatchMidpointVI(system, B)
    nX, nU = mvi.nX, mvi.nU
    R = mvi.nq + mvi.nd + mvi.nu + mvi.nk
    pool = _DevicePool(0)
    X = np.zeros((B, 2, nX)); X[:, 0, :mvi.nq] = Q0; X[:, 1, :mvi.nq] = Q0
    U = np.zeros((B, 1, nU)); U[:, 0, mvi.nu:] = K[:, 0]
    dX, dU = pool.upload(X), pool.upload(U)
    dA, dB = pool.empty((B, nX, nX)), pool.empty((B, nX, nU))
    dZ, dHZ = pool.upload(np.random.default_rng(0).standard_normal((B, nX))), pool.empty((B, R, R))
    out = {"batch": B, "system": "Puppet(string_constraints=True) nq=40 nd=22 nk=18 nc=6"}
    res = {}
    rng = np.random.default_rng(1)
    dQ, ddQ = pool.upload(Q0), pool.upload(rng.standard_normal((B, mvi.nq)))       # continuous dynamics inputs
    ddK, dACC, dLAM = pool.upload(rng.standard_normal((B, mvi.nk))), pool.empty((B, mvi.nd)), pool.empty((B, mvi.nc))
    import ctypes
    rows = (mvi.nq, mvi.nq, mvi.nk, mvi.nu)
    g1_bufs = [pool.empty((B, max(rows[g & 3], 1), mvi.nd if g < 4 else mvi.nc)) for g in range(8)]
    g1 = (ctypes.c_void_p * 8)(*[buf.ptr if rows[g & 3] else None for g, buf in enumerate(g1_bufs)])
    n_g1 = sum(rows[g & 3] * (mvi.nd if g < 4 else mvi.nc) for g in range(8))
    for rep in range(args.reps + 1):
        mvi.timing()
        _lib.check(L.tg_batch_set_from_trajectories(mvi._h, B, 1, 0.0, dt, dX.ptr, dU.ptr, 200))
        n, ms_step = mvi.timing()
        _lib.check(L.tg_batch_linearize(mvi._h, dA.ptr, dB.ptr))
        n, ms_d1 = mvi.timing()
        _lib.check(L.tg_batch_deriv2_contract_device(mvi._h, dZ.ptr, dHZ.ptr))
        n, ms_d2 = mvi.timing()
        _lib.check(L.tg_batch_dynamics_device(mvi._h, dQ.ptr, ddQ.ptr, None, ddK.ptr, dACC.ptr, dLAM.ptr, None))
        n, ms_dyn = mvi.timing()
        _lib.check(L.tg_batch_dynamics_deriv1_device(mvi._h, dQ.ptr, ddQ.ptr, None, ddK.ptr, g1, None))
        n, ms_dyn1 = mvi.timing()
        if rep:   # first pass = warm-up
            for k, v in (("step", ms_step), ("deriv1_AB", ms_d1), ("deriv2z", ms_d2), ("dynamics", ms_dyn), ("dynamics_deriv1", ms_dyn1)):
                res.setdefault(k, []).append(v)
    iters, status = mvi.status()
    assert (status == 0).all()
    bytes_per = {"step": 8 * (2 * nX + nU + mvi.nc), "deriv1_AB": 8 * (nX * nX + nX * nU), "deriv2z": 8 * (nX + R * R),
                 "dynamics": 8 * (2 * mvi.nq + mvi.nu + mvi.nk + mvi.nd + mvi.nc),
                 "dynamics_deriv1": 8 * (2 * mvi.nq + mvi.nu + mvi.nk + n_g1)}
    for k, v in res.items():
        ms = float(np.mean(v))
        out[k] = {"kernel_ms": ms, "per_s": B / ms * 1e3, "algorithmic_bytes_per_unit": bytes_per[k],
                  "achieved_GBps": B * bytes_per[k] / ms / 1e6, "hbm_peak_GBps": 8000.0}
    out["newton_iterations_per_step"] = float(iters.mean())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
