#!/usr/bin/env python3
"""Kernel time of MODE_DYN_DERIV1 on the puppet (library from TREPAMD_LIB if set; timing mocks allowed: the status is not checked).
  python tools/time_dyn_deriv1.py --batch 65536"""
import argparse, ctypes, json, os, sys
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
    B, nd = args.batch, system.nQd
    distinct = min(B, 256)
    Q0 = np.tile(systems.puppet_initial_conditions(system, distinct, seed=11), ((B + distinct - 1) // distinct, 1))[:B]
    mvi = trep_amd.BatchMidpointVI(system, B)
    pool = _DevicePool(0)
    rng = np.random.default_rng(1)
    dQ, ddQ, ddK = pool.upload(Q0), pool.upload(rng.standard_normal((B, mvi.nq))), pool.upload(rng.standard_normal((B, mvi.nk)))
    rows = (mvi.nq, mvi.nq, mvi.nk, mvi.nu)
    bufs = [pool.empty((B, max(rows[g & 3], 1), mvi.nd if g < 4 else mvi.nc)) for g in range(8)]
    g1 = (ctypes.c_void_p * 8)(*[buf.ptr if rows[g & 3] else None for g, buf in enumerate(bufs)])
    ms = []
    for rep in range(args.reps + 1):
        mvi.timing()
        _lib.check(L.tg_batch_dynamics_deriv1_device(mvi._h, dQ.ptr, ddQ.ptr, None, ddK.ptr, g1, None))
        n, t = mvi.timing()
        if rep:
            ms.append(t)
    print(json.dumps({"lib": os.environ.get("TREPAMD_LIB", "product"), "batch": B, "kernel_ms": float(np.mean(ms)), "per_s": B / float(np.mean(ms)) * 1e3}))


if __name__ == "__main__":
    main()
