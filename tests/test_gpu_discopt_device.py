"""Device-side discopt primitives (tg_tv_lq, tg_adjoint_sweep, tg_tangent_rollout, tg_quadratic_cost*, ...) and
the BatchDOptimizer built on them, against the host numpy implementation of the same formulas
(trep_amd.discopt.dlqr / DCost / DOptimizer, which tests/test_discopt.py pins to a DOptimizer trace recorded
from the reference).  Tolerance: 1e-9 relative for the Riccati outputs, 1e-10 for the rest."""
import ctypes

import numpy as np
import pytest

from common import golden, relerr

pytestmark = pytest.mark.gpu


def _pool():
    from trep_amd.discopt.batch_doptimizer import _DevicePool
    return _DevicePool(0)


def _random_problem(rng, S, N, nX, nU, nxh):
    A = 0.2 * rng.standard_normal((S, N, nX, nX)) / np.sqrt(nX) + 0.9 * np.eye(nX)
    B = rng.standard_normal((S, N, nX, nU)) / np.sqrt(nX)
    Q = rng.standard_normal((nX, nX)); Q = Q.dot(Q.T) / nX + np.eye(nX)
    Qf = 2.0 * Q
    R = rng.standard_normal((nU, nU)); R = R.dot(R.T) / nU + np.eye(nU)
    q = rng.standard_normal((S, N + 1, nX))
    r = rng.standard_normal((S, N, nU))
    Rz = nxh + nU
    hz = 0.05 * rng.standard_normal((S, N, Rz, Rz))
    hz = hz + np.swapaxes(hz, 2, 3)
    return A, B, Q, Qf, R, q, r, hz


def _host_lq(A, B, Q, Qf, R, q, r, hz, nxh):
    """numpy reference: dlqr.solve_tv_lq with the Newton-model weights assembled like DSystem._split_hz."""
    from trep_amd.discopt import dlqr
    N, nX, nU = A.shape[0], A.shape[1], B.shape[2]

    def Qk(k):
        if k == N:
            return Qf
        M = Q.copy()
        if hz is not None:
            M[:nxh, :nxh] += hz[k][:nxh, :nxh]
        return M

    def Sk(k):
        M = np.zeros((nX, nU))
        if hz is not None:
            M[:nxh, :] = hz[k][:nxh, nxh:]
        return M

    def Rk(k):
        return R + (hz[k][nxh:, nxh:] if hz is not None else 0.0)

    if q is None:
        assert hz is None
        K, P = dlqr.solve_tv_lqr(A, B, Qk, Rk)
        return np.array(K), None, P, None
    K, C, P, b = dlqr.solve_tv_lq(A, B, q, r, Qk, Sk, Rk)
    return np.array(K), np.array(C), P, b


@pytest.mark.parametrize("nX,nU,nxh,N,S", [(4, 1, 4, 40, 3), (18, 3, 12, 25, 2), (80, 18, 62, 30, 3), (90, 10, 70, 8, 1)])
def test_tv_lq_matches_numpy(nX, nU, nxh, N, S):
    from trep_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(100 + nX)
    A, B, Q, Qf, R, q, r, hz = _random_problem(rng, S, N, nX, nU, nxh)
    pool = _pool()
    try:
        dA, dB, dQ, dQf, dR = pool.upload(A), pool.upload(B), pool.upload(Q), pool.upload(Qf), pool.upload(R)
        dq, dr, dhz = pool.upload(q), pool.upload(r), pool.upload(hz)
        for mode in ("lqr", "lq", "newton", "subset"):
            dK, dC = pool.empty((S, N, nU, nX)).set(np.full((S, N, nU, nX), np.nan)), pool.empty((S, N, nU))
            dP, db, dst = pool.empty((S, nX, nX)), pool.empty((S, nX)), pool.empty((S,), np.int32)
            p = _lib.LqProblem()
            p.n_problems, p.horizon, p.nX, p.nU = S, N, nX, nU
            p.A_dev, p.B_dev = dA.ptr, dB.ptr
            p.Q_dev, p.Qf_dev, p.R_dev = dQ.ptr, dQf.ptr, dR.ptr
            p.K_dev, p.C_dev, p.P0_dev, p.b0_dev, p.status_dev = dK.ptr, dC.ptr, dP.ptr, db.ptr, dst.ptr
            seeds = list(range(S))
            if mode != "lqr":
                p.q_dev, p.r_dev = dq.ptr, dr.ptr
            if mode in ("newton", "subset"):
                p.hz_dev, p.hz_R, p.hz_nx = dhz.ptr, nxh + nU, nxh
            if mode == "subset":
                seeds = [S - 1]
                sel = pool.upload(np.array(seeds, dtype=np.int32), np.int32)
                p.select_dev, p.n_problems = sel.ptr, 1
            _lib.check(L.tg_tv_lq(0, ctypes.byref(p)))
            K, C, P0, b0, st = dK.get(), dC.get(), dP.get(), db.get(), dst.get()
            for s in range(S):
                if s not in seeds:
                    assert np.isnan(K[s]).all()       # untouched
                    continue
                assert st[s] == 0
                Kh, Ch, Ph, bh = _host_lq(A[s], B[s], Q, Qf, R, None if mode == "lqr" else q[s], None if mode == "lqr" else r[s],
                                          hz[s] if mode in ("newton", "subset") else None, nxh)
                assert relerr(K[s], Kh) < 1e-9, (mode, s, relerr(K[s], Kh))
                assert relerr(P0[s], Ph) < 1e-9, (mode, s)
                if mode != "lqr":
                    assert relerr(C[s], Ch) < 1e-9 and relerr(b0[s], bh) < 1e-9, (mode, s)
    finally:
        pool.close()


def _dsystem_structure(rng, A, B, nd, nk, nu, dt=0.01):
    """Impose the block structure of DSystem.fdx / fdu (dsystem.py:284-317) on random A, B: states [Qd | Qk | p | v], inputs [u | rho]."""
    nq, nX = nd + nk, 2 * (nd + nk)
    Qd, Qk, p, v = slice(0, nd), slice(nd, nq), slice(nq, nq + nd), slice(nq + nd, nX)
    A[..., Qk, :] = 0.0; A[..., v, :] = 0.0; A[..., :, v] = 0.0
    B[..., Qk, :] = 0.0; B[..., v, :] = 0.0
    for m in range(nk):
        steps = dt * (1.0 + 0.3 * rng.random(A.shape[:-2]))       # any time base: the entries are read, not assumed
        A[..., nq + nd + m, nd + m] = -1.0 / steps
        B[..., nd + m, nu + m] = 1.0
        B[..., nq + nd + m, nu + m] = 1.0 / steps
    return A, B


@pytest.mark.parametrize("nd,nk,nu,N,S", [(22, 18, 0, 30, 3), (9, 4, 2, 25, 2), (3, 5, 1, 20, 2), (30, 10, 0, 12, 1), (5, 0, 3, 20, 2)])
def test_tv_lq_with_dsystem_structure(nd, nk, nu, N, S):
    """k_tv_lq_mfma with tg_lq_problem::ds_* set: the products skip the zero blocks of DSystem.fdx / fdu (two k-ranges over the dense Qd / p
    rows, the single-entry Qk / v rows as extra terms, v columns never computed).  Against numpy's dlqr on the same matrices (1e-9) and
    against the dense sweep of the same kernel (1e-11): LQR, affine LQ and the Newton model with the on-the-fly curvature; the state
    weights couple every block (a dense Q_k), so P's v rows / columns are exercised too."""
    from trep_amd import _lib
    L = _lib.lib()
    nX, nU, nxh = 2 * (nd + nk), nu + nk, 2 * nd + nk
    rng = np.random.default_rng(500 + nd)
    A, B, Q, Qf, R, q, r, hz = _random_problem(rng, S, N, nX, nU, nxh)
    A, B = _dsystem_structure(rng, A, B, nd, nk, nu)
    pool = _pool()
    try:
        dA, dB, dQ, dQf, dR = pool.upload(A), pool.upload(B), pool.upload(Q), pool.upload(Qf), pool.upload(R)
        dq, dr, dhz = pool.upload(q), pool.upload(r), pool.upload(hz)
        for mode in ("lqr", "lq", "newton"):
            res = {}
            for structured in (True, False):
                dK, dC = pool.empty((S, N, nU, nX)), pool.empty((S, N, nU))
                dP, db, dst = pool.empty((S, nX, nX)), pool.empty((S, nX)), pool.empty((S,), np.int32)
                p = _lib.LqProblem()
                p.n_problems, p.horizon, p.nX, p.nU = S, N, nX, nU
                p.A_dev, p.B_dev = dA.ptr, dB.ptr
                p.Q_dev, p.Qf_dev, p.R_dev = dQ.ptr, dQf.ptr, dR.ptr
                p.K_dev, p.C_dev, p.P0_dev, p.b0_dev, p.status_dev = dK.ptr, dC.ptr, dP.ptr, db.ptr, dst.ptr
                if mode != "lqr":
                    p.q_dev, p.r_dev = dq.ptr, dr.ptr
                if mode == "newton":
                    p.hz_dev, p.hz_R, p.hz_nx = dhz.ptr, nxh + nU, nxh
                if structured:
                    p.ds_nd, p.ds_nk, p.ds_nu = nd, nk, nu
                _lib.check(L.tg_tv_lq(0, ctypes.byref(p)))
                assert (dst.get() == 0).all()
                res[structured] = (dK.get(), dC.get(), dP.get(), db.get())
            for s in range(S):
                Kh, Ch, Ph, bh = _host_lq(A[s], B[s], Q, Qf, R, None if mode == "lqr" else q[s], None if mode == "lqr" else r[s],
                                          hz[s] if mode == "newton" else None, nxh)
                K, C, P0, b0 = (x[s] for x in res[True])
                assert relerr(K, Kh) < 1e-9 and relerr(P0, Ph) < 1e-9, (mode, s, relerr(K, Kh), relerr(P0, Ph))
                assert relerr(K, res[False][0][s]) < 1e-11 and relerr(P0, res[False][2][s]) < 1e-11, (mode, s)
                assert not K[:, nX - nk:].any()       # the v columns of the gains are exactly zero
                if mode != "lqr":
                    assert relerr(C, Ch) < 1e-9 and relerr(b0, bh) < 1e-9, (mode, s)
                    assert relerr(C, res[False][1][s]) < 1e-11 and relerr(b0, res[False][3][s]) < 1e-11
    finally:
        pool.close()
    # sizes that contradict the structure are refused
    p = _lib.LqProblem()
    p.n_problems, p.horizon, p.nX, p.nU = 1, 1, nX, nU
    p.A_dev = p.B_dev = p.Q_dev = p.Qf_dev = p.R_dev = p.K_dev = 8
    p.ds_nd, p.ds_nk, p.ds_nu = nd + 1, nk, nu
    assert L.tg_tv_lq(0, ctypes.byref(p)) != 0


@pytest.mark.parametrize("kernel", ["structured", "dense", "legacy"])
def test_tv_lq_swept_in_chunks_is_the_same_sweep(kernel, monkeypatch):
    """tg_lq_problem::k_begin / k_end / Pt_dev / bt_dev: the horizon swept in three launches, each continuing from the (P, b) the previous one
    left through P0_dev / b0_dev, gives BIT FOR BIT the gains, affine terms, adjoint rows (b_next_dev) and (P_0, b_0) of one sweep -- for the
    DSystem-structured kernel, the dense matrix-core kernel and the VALU kernel -- and a sweep that ends before the horizon without the
    (P, b) behind it is refused.  (The pipelined Newton step of BatchDOptimizer rests on this.)"""
    from trep_amd import _lib
    L = _lib.lib()
    if kernel == "legacy":
        monkeypatch.setenv("TREPAMD_LQ_LEGACY", "1")
    nd, nk, nu, N, S = 22, 18, 0, 37, 3
    nX, nU, nxh = 2 * (nd + nk), nu + nk, 2 * nd + nk
    rng = np.random.default_rng(77)
    A, B, Q, Qf, R, q, r, hz = _random_problem(rng, S, N, nX, nU, nxh)
    A, B = _dsystem_structure(rng, A, B, nd, nk, nu)
    pool = _pool()
    try:
        dA, dB, dQ, dQf, dR = pool.upload(A), pool.upload(B), pool.upload(Q), pool.upload(Qf), pool.upload(R)
        dq, dr, dhz = pool.upload(q), pool.upload(r), pool.upload(hz)

        def problem(dK, dC, dZ, dst):
            p = _lib.LqProblem()
            p.n_problems, p.horizon, p.nX, p.nU = S, N, nX, nU
            p.A_dev, p.B_dev = dA.ptr, dB.ptr
            p.Q_dev, p.Qf_dev, p.R_dev = dQ.ptr, dQf.ptr, dR.ptr
            p.q_dev, p.r_dev = dq.ptr, dr.ptr
            p.hz_dev, p.hz_R, p.hz_nx = dhz.ptr, nxh + nU, nxh
            p.K_dev, p.C_dev, p.b_next_dev, p.status_dev = dK.ptr, dC.ptr, dZ.ptr, dst.ptr
            if kernel == "structured":
                p.ds_nd, p.ds_nk, p.ds_nu = nd, nk, nu
            return p
        outs = []
        for chunks in ([(0, N)], [(25, N), (9, 25), (0, 9)]):
            dK, dC, dZ, dst = pool.empty((S, N, nU, nX)), pool.empty((S, N, nU)), pool.empty((S, N, nX)), pool.empty((S,), np.int32)
            carry = [(pool.empty((S, nX, nX)), pool.empty((S, nX))) for _ in range(2)]
            for c, (k0, k1) in enumerate(chunks):
                p = problem(dK, dC, dZ, dst)
                p.P0_dev, p.b0_dev = carry[c % 2][0].ptr, carry[c % 2][1].ptr
                if len(chunks) > 1:
                    p.k_begin, p.k_end = k0, k1
                if c > 0:
                    p.Pt_dev, p.bt_dev = carry[(c - 1) % 2][0].ptr, carry[(c - 1) % 2][1].ptr
                _lib.check(L.tg_tv_lq(0, ctypes.byref(p)))
            last = carry[(len(chunks) - 1) % 2]
            assert (dst.get() == 0).all()
            outs.append((dK.get(), dC.get(), dZ.get(), last[0].get(), last[1].get()))
        for a, b in zip(*outs):
            assert np.array_equal(a, b)
        for s in range(S):      # ... and it is the right sweep
            Kh, Ch, Ph, bh = _host_lq(A[s], B[s], Q, Qf, R, q[s], r[s], hz[s], nxh)
            assert relerr(outs[1][0][s], Kh) < 1e-9 and relerr(outs[1][3][s], Ph) < 1e-9
        p = problem(dK, dC, dZ, dst)
        p.k_begin, p.k_end = 0, N - 3
        assert L.tg_tv_lq(0, ctypes.byref(p)) != 0
    finally:
        pool.close()


@pytest.mark.parametrize("nX,nU", [(6, 2), (30, 9), (37, 5), (80, 18), (91, 27), (96, 32)])
def test_tangent_rollout_sizes(nX, nU, monkeypatch):
    """tg_tangent_rollout (doptimizer.py:391-402, 262-270) against numpy over the slice sizes k_tangent_rows is compiled for (odd sizes: the
    clamped last columns meet the zero padding), with a selection list, and against the LDS-staged kernel it replaces."""
    from trep_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(100 * nX + nU)
    S, N = 3, 23
    A = rng.standard_normal((S, N, nX, nX)) * (0.9 / np.sqrt(nX)); B = rng.standard_normal((S, N, nX, nU)) * 0.3
    K = rng.standard_normal((S, N, nU, nX)) * 0.1; C = rng.standard_normal((S, N, nU))
    q = rng.standard_normal((S, N + 1, nX)); r = rng.standard_normal((S, N, nU))
    sel = np.array([2, 0], dtype=np.int32)
    pool = _pool()
    try:
        d = dict((k, pool.upload(v)) for k, v in dict(A=A, B=B, K=K, C=C, q=q, r=r).items())
        dsel = pool.upload(sel, np.int32)
        out = {}
        for variant in ("rows", "lds"):
            if variant == "lds":
                monkeypatch.setenv("TREPAMD_TANGENT_LDS", "1")
            ddX, ddU, ddc = pool.upload(np.full((S, N + 1, nX), np.nan)), pool.upload(np.full((S, N, nU), np.nan)), pool.upload(np.full((S,), np.nan))
            _lib.check(L.tg_tangent_rollout(0, len(sel), N, nX, nU, dsel.ptr, d["A"].ptr, d["B"].ptr, d["K"].ptr, d["C"].ptr, d["q"].ptr, d["r"].ptr,
                                            ddX.ptr, ddU.ptr, ddc.ptr))
            out[variant] = (ddX.get(), ddU.get(), ddc.get())
        dX, dU, dc = out["rows"]
        assert np.isnan(dX[1]).all() and np.isnan(dU[1]).all() and np.isnan(dc[1])      # not selected: untouched
        for s in sel:
            x = np.zeros(nX); ref = 0.0
            for k in range(N):
                assert relerr(dX[s, k], x) < 1e-11 or not x.any()
                u = -K[s, k].dot(x) - C[s, k]
                assert relerr(dU[s, k], u) < 1e-11
                ref += q[s, k].dot(x) + r[s, k].dot(u)
                x = A[s, k].dot(x) + B[s, k].dot(u)
            assert relerr(dX[s, N], x) < 1e-11
            ref += q[s, N].dot(x)
            assert abs(dc[s] - ref) < 1e-10 * max(1.0, abs(ref))
            assert relerr(dX[s], out["lds"][0][s]) < 1e-12 and relerr(dU[s], out["lds"][1][s]) < 1e-12
    finally:
        pool.close()


def test_sweeps_and_cost_match_numpy():
    from trep_amd import _lib
    from trep_amd.discopt import DCost
    L = _lib.lib()
    rng = np.random.default_rng(7)
    S, N, nX, nU, M = 3, 37, 80, 18, 4
    A, B, Q, Qf, R, q, r, _ = _random_problem(rng, S, N, nX, nU, 62)
    K = rng.standard_normal((S, N, nU, nX)) * 0.1
    C = rng.standard_normal((S, N, nU))
    X, U = rng.standard_normal((S, N + 1, nX)), rng.standard_normal((S, N, nU))
    Xd, Ud = rng.standard_normal((S, N + 1, nX)), rng.standard_normal((S, N, nU))
    pool = _pool()
    try:
        d = dict((k, pool.upload(v)) for k, v in dict(A=A, B=B, Q=Q, Qf=Qf, R=R, q=q, r=r, K=K, C=C, X=X, U=U, Xd=Xd, Ud=Ud).items())
        # adjoint (doptimizer.py:319-345)
        dZ = pool.empty((S, N, nX))
        _lib.check(L.tg_adjoint_sweep(0, S, N, nX, nU, None, d["A"].ptr, d["B"].ptr, d["K"].ptr, d["q"].ptr, d["r"].ptr, dZ.ptr))
        Z = dZ.get()
        for s in range(S):
            z = q[s, -1]
            for k in range(N - 1, -1, -1):
                assert relerr(Z[s, k], z) < 1e-11, (s, k)
                z = q[s, k] - r[s, k].dot(K[s, k]) + z.dot(A[s, k] - B[s, k].dot(K[s, k]))
        # tangent rollout + directional derivative (doptimizer.py:391-402, 262-270)
        ddX, ddU, ddc = pool.empty((S, N + 1, nX)), pool.empty((S, N, nU)), pool.empty((S,))
        _lib.check(L.tg_tangent_rollout(0, S, N, nX, nU, None, d["A"].ptr, d["B"].ptr, d["K"].ptr, d["C"].ptr, d["q"].ptr, d["r"].ptr,
                                        ddX.ptr, ddU.ptr, ddc.ptr))
        dX, dU, dc = ddX.get(), ddU.get(), ddc.get()
        for s in range(S):
            x = np.zeros(nX)
            for k in range(N):
                assert relerr(dX[s, k], x) < 1e-11
                u = -K[s, k].dot(x) - C[s, k]
                assert relerr(dU[s, k], u) < 1e-11
                x = A[s, k].dot(x) + B[s, k].dot(u)
            assert relerr(dX[s, N], x) < 1e-11
            ref = float(np.sum(q[s] * dX[s]) + np.sum(r[s] * dU[s]))
            assert abs(dc[s] - ref) < 1e-10 * max(1.0, abs(ref))
        # cost and gradients (dcost.py)
        dcost, dq2, dr2 = pool.empty((S,)), pool.empty((S, N + 1, nX)), pool.empty((S, N, nU))
        _lib.check(L.tg_quadratic_cost(0, S, 1, None, N, nX, nU, d["X"].ptr, d["U"].ptr, d["Xd"].ptr, d["Ud"].ptr, d["Q"].ptr, d["R"].ptr,
                                       d["Qf"].ptr, dcost.ptr))
        _lib.check(L.tg_quadratic_cost_gradients(0, S, N, nX, nU, None, d["X"].ptr, d["U"].ptr, d["Xd"].ptr, d["Ud"].ptr, d["Q"].ptr,
                                                 d["R"].ptr, d["Qf"].ptr, dq2.ptr, dr2.ptr))
        cost, gq, gr = dcost.get(), dq2.get(), dr2.get()
        for s in range(S):
            c = DCost(Xd[s], Ud[s], Q, R, Qf)
            assert abs(cost[s] - c.total(X[s], U[s])) < 1e-11 * abs(c.total(X[s], U[s]))
            hq, hr = c.gradients(X[s], U[s])
            assert relerr(gq[s], hq) < 1e-12 and relerr(gr[s], hr) < 1e-12
        # candidates and grouped cost, row copies
        lam = 0.7 ** np.arange(M)
        dlam, dbX, dbU = pool.upload(lam), pool.empty((S * M, N + 1, nX)), pool.empty((S * M, N, nU))
        _lib.check(L.tg_armijo_candidates(0, S, M, N, nX, nU, None, dlam.ptr, d["X"].ptr, d["U"].ptr, ddX.ptr, ddU.ptr, dbX.ptr, dbU.ptr))
        bX, bU = dbX.get().reshape(S, M, N + 1, nX), dbU.get().reshape(S, M, N, nU)
        assert relerr(bX, X[:, None] + lam[None, :, None, None] * dX[:, None]) < 1e-14   # the device contracts to an fma
        assert relerr(bU, U[:, None] + lam[None, :, None, None] * dU[:, None]) < 1e-14
        dcc = pool.empty((S * M,))
        _lib.check(L.tg_quadratic_cost(0, S * M, M, None, N, nX, nU, dbX.ptr, dbU.ptr, d["Xd"].ptr, d["Ud"].ptr, d["Q"].ptr, d["R"].ptr,
                                       d["Qf"].ptr, dcc.ptr))
        cc = dcc.get().reshape(S, M)
        for s in range(S):
            c = DCost(Xd[s], Ud[s], Q, R, Qf)
            assert relerr(cc[s], c.total(bX[s], bU[s])) < 1e-11
        rows_a = pool.upload(np.array([2, 0], dtype=np.int32), np.int32)
        rows_b = pool.upload(np.array([2 * M + 1, 0 * M + 3], dtype=np.int32), np.int32)
        _lib.check(L.tg_copy_rows(0, 2, (N + 1) * nX, rows_a.ptr, rows_b.ptr, dbX.ptr, d["X"].ptr))
        X2 = d["X"].get()
        assert np.array_equal(X2[2], bX[2, 1]) and np.array_equal(X2[0], bX[0, 3]) and np.array_equal(X2[1], X[1])
    finally:
        pool.close()


def _cart_problem(S):
    """Perturbed copies of the pend-on-cart problem of the reference trace (tests/golden/discopt_pend_on_cart.npz)."""
    from trep_amd import systems
    g = golden("discopt_pend_on_cart")
    system = systems.pend_on_cart(torque_force=True)
    rng = np.random.default_rng(5)
    Xd = np.repeat(g["Xd"][None], S, axis=0)
    Ud = np.repeat(g["Ud"][None], S, axis=0)
    Ud[1:] += 0.05 * rng.standard_normal(Ud[1:].shape)
    return g, system, Xd, Ud


def _sequential_steps(dsys, Xd, Ud, Q, R, X0, U0, methods):
    from trep_amd import discopt
    out = []
    for s in range(len(Xd)):
        opt = discopt.DOptimizer(dsys, discopt.DCost(Xd[s], Ud[s], Q, R))
        X, U = X0[s].copy(), U0[s].copy()
        rec = []
        for i, m in enumerate(methods):
            r = opt.step(i, X, U, m)
            rec.append((opt.monitor.cost_history[i], opt.monitor.dcost_history[i], r.cost1, r.nX.copy(), r.nU.copy()))
            X, U = r.nX, r.nU
        out.append(rec)
    return out


def test_batch_optimizer_matches_sequential_cart():
    import trep_amd
    from trep_amd import discopt
    S = 5
    g, system, Xd, Ud = _cart_problem(S)
    t = g["t"]
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), t)
    X0 = np.repeat(g["X0"][None], S, axis=0)
    U0 = np.repeat(g["U0"][None], S, axis=0)
    methods = ["quasi", "quasi", "newton", "newton"]
    ref = _sequential_steps(dsys, Xd, Ud, g["Q"], g["R"], X0, U0, methods)
    opt = discopt.BatchDOptimizer(dsys, Xd, Ud, g["Q"], g["R"], armijo_chunk=2)
    try:
        opt.set_trajectories(X0, U0)
        for i, m in enumerate(methods):
            r = opt.step(m)
            X, U = opt.get_trajectories()
            for s in range(S):
                c0, dc0, c1, Xr, Ur = ref[s][i]
                assert abs(r.cost0[s] - c0) < 1e-9 * max(1.0, abs(c0)), (i, s)
                assert abs(r.dcost0[s] - dc0) < 1e-7 * max(1.0, abs(dc0)), (i, s, r.dcost0[s], dc0)
                assert abs(r.cost1[s] - c1) < 1e-8 * max(1.0, abs(c1)), (i, s, r.cost1[s], c1)
                assert relerr(X[s], Xr) < 1e-7 and relerr(U[s], Ur) < 1e-7, (i, s, relerr(X[s], Xr))
        # a demanding sufficient-decrease constant forces m > 0: exercises the chunked search (2 candidates per launch)
        from trep_amd import discopt as _d
        Xs, Us = opt.get_trajectories()
        opt.armijo_alpha = 0.7
        r = opt.step("quasi")
        X, U = opt.get_trajectories()
        assert (r.armijo > 0).any() and not r.failed.any()
        for s in range(S):
            o = _d.DOptimizer(dsys, _d.DCost(Xd[s], Ud[s], g["Q"], g["R"]))
            o.armijo_alpha = 0.7
            m_seen = []
            o.monitor.armijo_evaluation = lambda m, *a: m_seen.append(m)
            rr = o.step(0, Xs[s], Us[s], "quasi")
            assert m_seen[-1] == r.armijo[s], (s, m_seen, r.armijo[s])
            assert abs(rr.cost1 - r.cost1[s]) < 1e-8 * max(1.0, abs(rr.cost1))
            assert relerr(X[s], rr.nX) < 1e-7
    finally:
        opt.close()


def test_adjoint_comes_out_of_the_projection_sweep():
    """projection_gain(with_adjoint=True): the Riccati sweep with the cost gradients as affine terms writes Z[s][k] = z_{k+1}
    (doptimizer.py:340-343) -- against the separate backward sweep tg_adjoint_sweep; the gain itself is bit-equal either way."""
    import trep_amd
    from trep_amd import discopt
    S = 3
    g, system, Xd, Ud = _cart_problem(S)
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])
    opt = discopt.BatchDOptimizer(dsys, Xd, Ud, g["Q"], g["R"], armijo_chunk=2)
    try:
        opt.set_trajectories(np.repeat(g["X0"][None], S, axis=0), np.repeat(g["U0"][None], S, axis=0))
        opt.linearize()
        opt.gradients_and_cost()
        opt.projection_gain()
        K0 = opt.Kproj.get()
        opt.newton_curvature(None)            # tg_adjoint_sweep
        Z0 = opt.Z.get()
        opt.projection_gain(with_adjoint=True)
        assert opt._adjoint_ready and np.array_equal(opt.Kproj.get(), K0)
        Z1 = opt.Z.get()
        assert relerr(Z1, Z0) < 1e-11, relerr(Z1, Z0)
        assert np.abs(Z0).max() > 0
    finally:
        opt.close()


def test_side_by_side_sweeps_change_nothing():
    """overlap_sweeps: projection gain and quasi-Newton sweep on two streams (default with few seeds) against one after the
    other; pipeline_newton: the projection sweep, the second derivatives and the Newton-model sweep chunk by chunk of the horizon
    in three stream lanes (tg_lq_problem::k_begin / Pt_dev) against one after the other -- same kernels on the same inputs, so
    every number of every step is bit-equal, including a step in which some seeds fall back from the Newton to the quasi-Newton
    direction (taken from the side-by-side sweep's buffers)."""
    import trep_amd
    from trep_amd import discopt
    S = 5
    g, system, Xd, Ud = _cart_problem(S)
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])
    X0 = np.repeat(g["X0"][None], S, axis=0)
    U0 = np.repeat(g["U0"][None], S, axis=0)
    trace = []
    for overlap, pipeline in ((False, False), (True, False), (True, True)):
        opt = discopt.BatchDOptimizer(dsys, Xd, Ud, g["Q"], g["R"], armijo_chunk=2, overlap_sweeps=overlap, pipeline_newton=pipeline)
        assert opt.overlap == overlap and opt.pipeline == pipeline
        opt.pipeline_chunks = 5
        try:
            opt.set_trajectories(X0, U0)
            out = []
            for m in ["quasi", "newton", ["newton", "quasi", "steepest", "newton", "quasi"]]:
                r = opt.step(m)
                out.append((r.cost0, r.dcost0, r.cost1, r.armijo, list(r.method)) + opt.get_trajectories())
            if pipeline:
                assert len(opt._chunks()) > 1
            # a Newton step whose model is forced indefinite for two seeds: they fall back to the quasi direction
            real, real_take = opt.descent_direction, opt._take_newton_direction
            def sabotaged(seeds, method):
                real(seeds, method)
                if method == "newton":
                    dc = opt.dcost.get(); dc[[1, 3]] = 1.0; opt.dcost.set(dc)
            def sabotaged_take(seeds):
                dc = np.array(real_take(seeds))
                idx = np.arange(S) if seeds is None else np.asarray(seeds)
                dc[np.isin(idx, [1, 3])] = 1.0
                return dc
            opt.descent_direction = sabotaged
            opt._take_newton_direction = sabotaged_take
            r = opt.step("newton")
            assert list(r.method) == ["newton", "quasi", "newton", "quasi", "newton"] and not r.failed.any()
            out.append((r.cost0, r.dcost0, r.cost1, r.armijo, list(r.method)) + opt.get_trajectories())
            trace.append(out)
        finally:
            opt.close()
    for other in trace[1:]:
        for a, b in zip(trace[0], other):
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x), np.asarray(y))


def test_armijo_speculation_depth_changes_nothing():
    """The first Armijo round only speculates twice as far as the last step with the same method needed (BatchDOptimizer.step): which
    candidate every seed accepts, its cost and its trajectory are those of the full-depth search, bit for bit -- also when a seed needs
    more candidates than the shortened first round holds."""
    import trep_amd
    from trep_amd import discopt
    S = 5
    g, system, Xd, Ud = _cart_problem(S)
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])
    X0 = np.repeat(g["X0"][None], S, axis=0)
    U0 = np.repeat(g["U0"][None], S, axis=0)
    trace, depths = [], []
    for adaptive in (False, True):
        opt = discopt.BatchDOptimizer(dsys, Xd, Ud, g["Q"], g["R"], armijo_chunk=12)
        real = opt.armijo_chunk
        seen = []
        def spy(m0, seeds=None, count=None, real=real, seen=seen):
            seen.append((m0, count))
            return real(m0, seeds, count)
        opt.armijo_chunk = spy
        try:
            opt.set_trajectories(X0, U0)
            out = []
            for i, m in enumerate(["quasi", "quasi", "newton", "newton", "quasi", "newton"]):
                if not adaptive:
                    opt._armijo_hint.clear()
                elif i == 4:
                    opt._armijo_hint["quasi"] = 0        # a hint that is too optimistic: the first round holds 4 candidates
                r = opt.step(m)
                assert not r.failed.any()
                out.append((r.cost0, r.dcost0, r.cost1, r.armijo) + opt.get_trajectories())
            trace.append(out)
            depths.append(seen)
        finally:
            opt.close()
    for a, b in zip(*trace):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    assert all(count == 12 for m0, count in depths[0] if m0 == 0)
    assert any(count < 12 for m0, count in depths[1] if m0 == 0)


def test_batch_optimizer_matches_sequential_puppet():
    import trep_amd
    from trep_amd import systems, discopt
    S, N, dt = 3, 40, 0.01
    system = systems.puppet()
    nd = system.nQd
    t = dt * np.arange(N + 1)
    Q0 = systems.puppet_initial_conditions(system, S, seed=77)
    K_move = systems.puppet_string_schedule(system, Q0[:, nd:], N, dt)
    K_still = np.repeat(Q0[:, None, nd:], N, axis=1)
    sim = trep_amd.BatchMidpointVI(system, S)
    sim.initialize_from_state(0.0, Q0, np.zeros((S, nd)))
    Xd = sim.rollout(N, dt, None, K_move)
    sim.initialize_from_state(0.0, Q0, np.zeros((S, nd)))
    Xi = sim.rollout(N, dt, None, K_still)
    sim.close()
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), t)
    wq = [100.0] * nd + [1.0] * system.nQk + [1.0] * nd + [1.0] * system.nQk
    Qc, Rc = np.diag(wq), np.diag([0.1] * system.nQk)
    methods = ["quasi", "newton"]
    ref = _sequential_steps(dsys, Xd, K_move, Qc, Rc, Xi, K_still, methods)
    opt = discopt.BatchDOptimizer(dsys, Xd, K_move, Qc, Rc)
    try:
        opt.set_trajectories(Xi, K_still)
        for i, m in enumerate(methods):
            r = opt.step(m)
            X, U = opt.get_trajectories()
            for s in range(S):
                c0, dc0, c1, Xr, Ur = ref[s][i]
                assert abs(r.cost0[s] - c0) < 1e-9 * max(1.0, abs(c0)), (i, s)
                assert abs(r.dcost0[s] - dc0) < 1e-7 * max(1.0, abs(dc0)), (i, s, r.dcost0[s], dc0)
                assert abs(r.cost1[s] - c1) < 1e-8 * max(1.0, abs(c1)), (i, s, r.cost1[s], c1)
                assert relerr(X[s], Xr) < 1e-7 and relerr(U[s], Ur) < 1e-7, (i, s, relerr(X[s], Xr))
    finally:
        opt.close()


def test_batch_optimizer_matches_sequential_wrench_arm():
    """Inputs that enter through HybridWrench forces and a kinematic config: the Newton steps exercise the input blocks
    of the contracted second derivatives (D1D3fm2 / D2D3fm2) inside the device-resident optimiser."""
    import trep_amd
    from trep_amd import discopt, systems
    from common import golden as gold
    g = gold("wrench_arm")
    system = systems.wrench_arm()
    N, S, dt = 60, 3, 0.01
    t = dt * np.arange(N + 1)
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), t)
    X_all, U_all = g["ds_X"], g["ds_U"]                      # a true trajectory of the reference rollout
    rng = np.random.default_rng(4)
    Xd = np.repeat(X_all[None, :N + 1], S, axis=0)
    Ud = np.repeat(U_all[None, :N], S, axis=0)
    X0, U0 = [], []
    for s in range(S):                                        # initial guesses: projections of perturbed copies
        bX = Xd[s] + 0.02 * rng.standard_normal(Xd[s].shape)
        bU = Ud[s] + 0.2 * rng.standard_normal(Ud[s].shape)
        bX[0] = Xd[s][0]
        pX, pU = dsys.project(bX, bU)
        X0.append(pX); U0.append(pU)
    X0, U0 = np.array(X0), np.array(U0)
    Q = np.diag(rng.uniform(0.5, 2.0, dsys.nX))
    R = np.diag(rng.uniform(0.05, 0.2, dsys.nU))
    methods = ["quasi", "newton", "newton"]
    ref = _sequential_steps(dsys, Xd, Ud, Q, R, X0, U0, methods)
    opt = discopt.BatchDOptimizer(dsys, Xd, Ud, Q, R, armijo_chunk=4)
    try:
        opt.set_trajectories(X0, U0)
        for i, m in enumerate(methods):
            r = opt.step(m)
            X, U = opt.get_trajectories()
            for s in range(S):
                c0, dc0, c1, Xr, Ur = ref[s][i]
                assert abs(r.cost0[s] - c0) < 1e-9 * max(1.0, abs(c0)), (i, s)
                assert abs(r.dcost0[s] - dc0) < 1e-6 * max(1.0, abs(dc0)), (i, s, r.dcost0[s], dc0)
                assert abs(r.cost1[s] - c1) < 1e-7 * max(1.0, abs(c1)), (i, s, r.cost1[s], c1)
                assert relerr(X[s], Xr) < 1e-6 and relerr(U[s], Ur) < 1e-6, (i, s, relerr(X[s], Xr))
    finally:
        opt.close()
