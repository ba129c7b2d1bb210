#!/usr/bin/env python3
"""Wall time of the device Riccati (solve_tv_lqr) and LQ (solve_tv_lq) sweeps at the puppet's size for several seed
counts; TREPAMD_LQ_LEGACY=1 selects the VALU kernel.  Prints one JSON line."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trep_amd import _lib
from trep_amd.discopt.batch_doptimizer import _DevicePool
nX, nU, N = 80, 18, 200
L = _lib.lib()
rng = np.random.default_rng(0)
out = {"kernel": "legacy VALU" if os.environ.get("TREPAMD_LQ_LEGACY") == "1" else "mfma", "nX": nX, "nU": nU, "N": N}
for S in (32, 256):
    A = 0.2 * rng.standard_normal((S, N, nX, nX)) / np.sqrt(nX) + 0.9 * np.eye(nX)
    B = rng.standard_normal((S, N, nX, nU)) / np.sqrt(nX)
    pool = _DevicePool(0)
    dA, dB, dQ, dR = pool.upload(A), pool.upload(B), pool.upload(np.eye(nX)), pool.upload(np.eye(nU))
    dq, dr = pool.upload(rng.standard_normal((S, N + 1, nX))), pool.upload(rng.standard_normal((S, N, nU)))
    dK, dC = pool.empty((S, N, nU, nX)), pool.empty((S, N, nU))
    # the block structure of DSystem.fdx / fdu (puppet: 22 dynamic + 18 kinematic configs): zero Qk rows / v columns of A, single-entry rows
    nd, nk = 22, 18
    nq = nd + nk
    A[:, :, nd:nq, :] = 0.0; A[:, :, nq + nd:, :] = 0.0; A[:, :, :, nq + nd:] = 0.0
    B[:, :, nd:nq, :] = 0.0; B[:, :, nq + nd:, :] = 0.0
    for m in range(nk):
        A[:, :, nq + nd + m, nd + m] = -100.0; B[:, :, nd + m, m] = 1.0; B[:, :, nq + nd + m, m] = 100.0
    dA.set(A); dB.set(B)
    nxh = 2 * nd + nk
    hz = 0.02 * rng.standard_normal((S, N, nxh + nU, nxh + nU)); hz = hz + np.swapaxes(hz, 2, 3)
    dhz = pool.upload(hz)
    for affine, structured, newton in ((False, False, False), (True, False, False), (False, True, False), (True, True, False), (True, True, True), (True, False, True)):
        p = _lib.LqProblem()
        if structured:
            p.ds_nd, p.ds_nk, p.ds_nu = nd, nk, 0
        if newton:      # the Newton model: curvature blocks read from HZ on the fly
            p.hz_dev, p.hz_R, p.hz_nx = dhz.ptr, nxh + nU, nxh
        p.n_problems, p.horizon, p.nX, p.nU = S, N, nX, nU
        p.A_dev, p.B_dev, p.Q_dev, p.Qf_dev, p.R_dev, p.K_dev, p.C_dev = dA.ptr, dB.ptr, dQ.ptr, dQ.ptr, dR.ptr, dK.ptr, dC.ptr
        if affine:
            p.q_dev, p.r_dev = dq.ptr, dr.ptr
        best = 1e9
        for rep in range(3):
            L.tg_device_synchronize(0)
            t0 = time.perf_counter()
            _lib.check(L.tg_tv_lq(0, ctypes.byref(p)))
            L.tg_device_synchronize(0)
            best = min(best, time.perf_counter() - t0)
        out["S%d_%s%s%s_us_per_k" % (S, "lq" if affine else "lqr", "_newton_model" if newton else "", "_dsystem" if structured else "")] = round(best / N * 1e6, 2)
    pool.close()
print(json.dumps(out))
