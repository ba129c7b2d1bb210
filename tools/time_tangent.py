#!/usr/bin/env python3
"""Kernel time of tg_tangent_rollout at the puppet's sizes (library from TREPAMD_LIB if set; timing mocks allowed).
  python tools/time_tangent.py --seeds 32 --horizon 1000"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=32)
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--nx", type=int, default=80)
    ap.add_argument("--nu", type=int, default=18)
    args = ap.parse_args()
    from trep_amd import _lib
    from trep_amd.discopt.batch_doptimizer import _DevicePool
    L = _lib.lib()
    S, N, nX, nU = args.seeds, args.horizon, args.nx, args.nu
    rng = np.random.default_rng(0)
    pool = _DevicePool(0)
    A = pool.upload(0.1 * rng.standard_normal((S, N, nX, nX)) / np.sqrt(nX)); B = pool.upload(0.1 * rng.standard_normal((S, N, nX, nU)))
    K = pool.upload(0.1 * rng.standard_normal((S, N, nU, nX))); C = pool.upload(rng.standard_normal((S, N, nU)))
    q = pool.upload(rng.standard_normal((S, N + 1, nX))); r = pool.upload(rng.standard_normal((S, N, nU)))
    dX, dU, dc = pool.empty((S, N + 1, nX)), pool.empty((S, N, nU)), pool.empty((S,))
    ms = []
    for rep in range(4):
        _lib.check(L.tg_device_synchronize(0))
        t0 = time.perf_counter()
        _lib.check(L.tg_tangent_rollout(0, S, N, nX, nU, None, A.ptr, B.ptr, K.ptr, C.ptr, q.ptr, r.ptr, dX.ptr, dU.ptr, dc.ptr))
        _lib.check(L.tg_device_synchronize(0))
        ms.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"lib": os.environ.get("TREPAMD_LIB", "product"), "seeds": S, "horizon": N, "ms": min(ms[1:]), "us_per_step": min(ms[1:]) / N * 1e3}))


if __name__ == "__main__":
    main()
