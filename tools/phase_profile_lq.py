#!/usr/bin/env python3
"""Cycle shares inside the LDS-resident Riccati / LQ sweep (diagnostic build):
TREPAMD_LIB=trep_amd/libtrepamd_prof.so python tools/phase_profile_lq.py [nX nU N]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trep_amd import _lib
from trep_amd.discopt.batch_doptimizer import _DevicePool
nX, nU, N, S = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) + (256,) if len(sys.argv) > 3 else (80, 18, 200, 256)
L = _lib.lib()
rng = np.random.default_rng(0)
A = 0.2 * rng.standard_normal((S, N, nX, nX)) / np.sqrt(nX) + 0.9 * np.eye(nX)
B = rng.standard_normal((S, N, nX, nU)) / np.sqrt(nX)
pool = _DevicePool(0)
dA, dB, dQ, dR = pool.upload(A), pool.upload(B), pool.upload(np.eye(nX)), pool.upload(np.eye(nU))
dq, dr = pool.upload(rng.standard_normal((S, N + 1, nX))), pool.upload(rng.standard_normal((S, N, nU)))
dK, dC = pool.empty((S, N, nU, nX)), pool.empty((S, N, nU))
nd, nk = 22, 18
if nX == 80 and nU == 18:      # the puppet's DSystem block structure (dsystem.py:284-317)
    nq = nd + nk
    A[:, :, nd:nq, :] = 0.0; A[:, :, nq + nd:, :] = 0.0; A[:, :, :, nq + nd:] = 0.0
    B[:, :, nd:nq, :] = 0.0; B[:, :, nq + nd:, :] = 0.0
    for m in range(nk):
        A[:, :, nq + nd + m, nd + m] = -100.0; B[:, :, nd + m, m] = 1.0; B[:, :, nq + nd + m, m] = 100.0
    dA.set(A); dB.set(B)
for affine, structured in ((False, False), (True, False), (False, True), (True, True)):
    if structured and not (nX == 80 and nU == 18):
        continue
    p = _lib.LqProblem()
    if structured:
        p.ds_nd, p.ds_nk, p.ds_nu = nd, nk, 0
    p.n_problems, p.horizon, p.nX, p.nU = S, N, nX, nU
    p.A_dev, p.B_dev, p.Q_dev, p.Qf_dev, p.R_dev, p.K_dev, p.C_dev = dA.ptr, dB.ptr, dQ.ptr, dQ.ptr, dR.ptr, dK.ptr, dC.ptr
    if affine:
        p.q_dev, p.r_dev = dq.ptr, dr.ptr
    L.tg_device_synchronize(0)
    t0 = time.perf_counter()
    _lib.check(L.tg_tv_lq(0, ctypes.byref(p)))
    L.tg_device_synchronize(0)
    el = time.perf_counter() - t0
    out = (ctypes.c_int64 * 8)()
    L.tg_lq_profile.argtypes = [ctypes.c_int32, ctypes.c_void_p]
    _lib.check(L.tg_lq_profile(0, out))
    v = np.array(list(out)[:8], dtype=float)
    print("%s nX=%d nU=%d N=%d S=%d: %.1f ms, %.1f us per k; cycles per k %.0f" % (("LQ (affine)" if affine else "LQR") + (" with the DSystem structure" if structured else " dense"), nX, nU, N, S, el * 1e3, el / N * 1e6, v.sum() / N))
    for n_, c in zip(["P.A tile, B'P, B'b", "PA->LDS, gamma, Kpart", "Gauss-Jordan (nU x nU)", "K out, new P tile (A'PA - Kpart'K), new b", "P, b, next A/B into LDS", "symmetrise", "(mfma kernel) solve of wave 0 without the barrier", "(DSystem kernel) phase 3 barrier wait (factorisation vs tiles), before the replay"], v):
        print("  %-44s %10.0f /k  %5.1f%%" % (n_, c / N, 100 * c / v.sum()))
