"""DSystem: the variational integrator seen as a discrete system X[k+1] = f(X[k], U[k], k).

Mirror of the reference's ``trep.discopt.DSystem`` (/root/reference/trep/discopt/dsystem.py:11-423)
for the part that sits directly on the MidpointVI hot path: state/input packing
  X[k] = [Q[k]; p[k]; v[k]],  v[k] = (rho[k] - rho[k-1]) / (t[k] - t[k-1]),  U[k] = [u[k]; rho[k+1]],
``set`` / ``step`` / ``f`` / ``fdx`` / ``fdu`` / ``linearize_trajectory``.  The integrator underneath
is the HIP one.  Second-order terms (fdxdx, fdxdu, fdudu) come from the z-contracted deriv2 kernel.

``BatchDSystem`` is the batched counterpart: B trajectories share the time base; ``linearize``
returns A [B][nX][nX], B [B][nX][nU] for every trajectory of the batch from one deriv1 launch.
"""
from collections import namedtuple

import numpy as np

from ..midpointvi import BatchMidpointVI


class _Packing(object):
    def _setup_packing(self, system, t):
        self._time = np.array(t, dtype=float).squeeze()
        self._nQ = len(system.configs)
        self._np = len(system.dyn_configs)
        self._nv = len(system.kin_configs)
        self._nu = len(system.inputs)
        self._nrho = len(system.kin_configs)
        self._nX = self._nQ + self._np + self._nv
        self._nU = self._nu + self._nrho
        self._slice_Q = slice(0, self._nQ)
        self._slice_Qd = slice(0, self._np)
        self._slice_Qk = slice(self._np, self._nQ)
        self._slice_p = slice(self._nQ, self._nQ + self._np)
        self._slice_v = slice(self._nQ + self._np, self._nX)
        self._slice_u = slice(0, self._nu)
        self._slice_rho = slice(self._nu, self._nU)
        self.trajectory_return = namedtuple('trajectory', 'X U')
        self.split_state_return = namedtuple('split_state', 'Q p v')
        self.split_input_return = namedtuple('split_input', 'u rho')
        self.split_trajectory_return = namedtuple('split_trajectory', 'Q p v u rho')
        self.linearization_return = namedtuple('linearization', 'A B')
        self.tangent_trajectory_return = namedtuple('tangent_trajectory', 'dX dU')
        self.feedback_return = namedtuple('feedback', 'Kproj A B')
        self.error_return = namedtuple('error', 'error exact_norm approx_norm')

    nX = property(lambda self: self._nX)
    nU = property(lambda self: self._nU)

    @property
    def time(self):
        return self._time

    @time.setter
    def time(self, t):
        t = np.array(t, dtype=float).squeeze()
        assert t.ndim == 1
        self._time = t

    def kf(self):
        return len(self._time) - 1

    def build_state(self, Q=None, p=None, v=None):
        X = np.zeros((self._nX,))
        if Q is not None:
            X[self._slice_Q] = Q
        if p is not None:
            X[self._slice_p] = p
        if v is not None:
            X[self._slice_v] = v
        return X

    def build_input(self, u=None, rho=None):
        U = np.zeros((self._nU,))
        if u is not None:
            U[self._slice_u] = u
        if rho is not None:
            U[self._slice_rho] = rho
        return U

    def build_trajectory(self, Q=None, p=None, v=None, u=None, rho=None):
        n = len(self._time)
        for name, val, want in (('Q', Q, n), ('p', p, n), ('v', v, n), ('u', u, n - 1), ('rho', rho, n - 1)):
            if val is not None and len(val) != want:
                raise Exception("Invalid length for %s (expected %d)" % (name, want))
        X = np.zeros((n, self._nX))
        U = np.zeros((n - 1, self._nU))
        if Q is not None:
            X[:, self._slice_Q] = Q
        if p is not None:
            X[:, self._slice_p] = p
        if v is not None:
            X[:, self._slice_v] = v
        if u is not None:
            U[:, self._slice_u] = u
        if rho is not None:
            U[:, self._slice_rho] = rho
        return self.trajectory_return(X, U)

    def split_state(self, X=None):
        if X is None:
            X = np.zeros(self._nX)
        return self.split_state_return(X[..., self._slice_Q], X[..., self._slice_p], X[..., self._slice_v])

    def split_input(self, U=None):
        if U is None:
            U = np.zeros(self._nU)
        return self.split_input_return(U[..., self._slice_u], U[..., self._slice_rho])

    def split_trajectory(self, X=None, U=None):
        if X is None and U is None:
            X = np.zeros((len(self._time), self._nX))
            U = np.zeros((len(self._time) - 1, self._nU))
        elif X is None:
            X = np.zeros((U.shape[0] + 1, self._nX))
        elif U is None:
            U = np.zeros((X.shape[0] - 1, self._nU))
        return self.split_trajectory_return(X[:, self._slice_Q], X[:, self._slice_p], X[:, self._slice_v],
                                            U[:, self._slice_u], U[:, self._slice_rho])

    def _split_hz(self, hz):
        """(fdxdx, fdxdu, fdudu) from HZ [R][R], variables ordered (q1, p1, u1, k2) (dsystem.py:320-386)."""
        nQ, npp, nu, nk = self._nQ, self._np, self._nu, self._nrho
        q1 = slice(0, nQ); p1 = slice(nQ, nQ + npp); u1 = slice(nQ + npp, nQ + npp + nu); k2 = slice(nQ + npp + nu, nQ + npp + nu + nk)
        xx = np.zeros((self._nX, self._nX)); xu = np.zeros((self._nX, self._nU)); uu = np.zeros((self._nU, self._nU))
        xx[self._slice_Q, self._slice_Q] = hz[q1, q1]
        xx[self._slice_Q, self._slice_p] = hz[q1, p1]
        xx[self._slice_p, self._slice_Q] = hz[p1, q1]
        xx[self._slice_p, self._slice_p] = hz[p1, p1]
        xu[self._slice_Q, self._slice_u] = hz[q1, u1]
        xu[self._slice_Q, self._slice_rho] = hz[q1, k2]
        xu[self._slice_p, self._slice_u] = hz[p1, u1]
        xu[self._slice_p, self._slice_rho] = hz[p1, k2]
        uu[self._slice_u, self._slice_u] = hz[u1, u1]
        uu[self._slice_u, self._slice_rho] = hz[u1, k2]
        uu[self._slice_rho, self._slice_u] = hz[k2, u1]
        uu[self._slice_rho, self._slice_rho] = hz[k2, k2]
        return xx, xu, uu

    def _assemble(self, dt, q2_dq1, q2_dp1, q2_du1, q2_dk2, p2_dq1, p2_dp1, p2_du1, p2_dk2):
        """A = fdx, B = fdu from output-major derivative blocks (dsystem.py:284-317)."""
        A = np.zeros((self._nX, self._nX))
        A[self._slice_Qd, self._slice_Q] = q2_dq1
        A[self._slice_Qd, self._slice_p] = q2_dp1
        A[self._slice_p, self._slice_Q] = p2_dq1
        A[self._slice_p, self._slice_p] = p2_dp1
        A[self._slice_v, self._slice_Qk] = np.diag(np.ones(self._nv) * -1.0 / dt)
        Bm = np.zeros((self._nX, self._nU))
        Bm[self._slice_Qd, self._slice_u] = q2_du1
        Bm[self._slice_Qd, self._slice_rho] = q2_dk2
        Bm[self._slice_Qk, self._slice_rho] = np.eye(self._nrho)
        Bm[self._slice_p, self._slice_u] = p2_du1
        Bm[self._slice_p, self._slice_rho] = p2_dk2
        Bm[self._slice_v, self._slice_rho] = np.diag(np.ones(self._nrho) * 1.0 / dt)
        return A, Bm


class DSystem(_Packing):
    def __init__(self, varint, t):
        self.varint = varint
        self._xk = None
        self._uk = None
        self._k = None
        self._setup_packing(varint.system, t)

    system = property(lambda self: self.varint.system)
    xk = property(lambda self: self._xk.copy())
    uk = property(lambda self: self._uk.copy())
    k = property(lambda self: self._k)

    def set(self, xk, uk, k, xk_hint=None, lambda_hint=None):
        self._k = k
        self._xk = xk.copy()
        self._uk = uk.copy()
        (q1, p1, v1) = self.split_state(xk)
        (u1, rho2) = self.split_input(uk)
        t1 = self._time[self._k + 0]
        t2 = self._time[self._k + 1]
        q2_hint = None
        if xk_hint is not None:
            q2_hint = self.split_state(xk_hint)[0][:self.varint.nd]
        self.varint.initialize_from_state(t1, q1, p1)
        self.varint.step(t2, u1, rho2, q2_hint=q2_hint, lambda1_hint=lambda_hint)

    def step(self, uk, xk_hint=None, lambda_hint=None):
        self._xk = self.f()
        self._uk = uk.copy()
        self._k += 1
        (u1, rho2) = self.split_input(uk)
        t2 = self._time[self._k + 1]
        q2_hint = None
        if xk_hint is not None:
            q2_hint = self.split_state(xk_hint)[0][:self.varint.nd]
        self.varint.step(t2, u1, rho2, q2_hint=q2_hint, lambda1_hint=lambda_hint)

    def f(self):
        return self.build_state(self.varint.q2, self.varint.p2, self.varint.v2)

    def _dt(self):
        return self._time[self._k + 1] - self._time[self._k + 0]

    def fdx(self):
        v = self.varint
        return self._assemble(self._dt(), v.q2_dq1(), v.q2_dp1(), v.q2_du1(), v.q2_dk2(),
                              v.p2_dq1(), v.p2_dp1(), v.p2_du1(), v.p2_dk2())[0]

    def fdu(self):
        v = self.varint
        return self._assemble(self._dt(), v.q2_dq1(), v.q2_dp1(), v.q2_du1(), v.q2_dk2(),
                              v.p2_dq1(), v.p2_dp1(), v.p2_du1(), v.p2_dk2())[1]

    def _second_order(self, z):
        hz = self.varint.deriv2_contract(z)
        return self._split_hz(hz)

    def fdxdx(self, z):
        """Second derivative of f w.r.t. the state, outputs contracted with z (dsystem.py:320-338)."""
        return self._second_order(z)[0]

    def fdxdu(self, z):
        return self._second_order(z)[1]

    def fdudu(self, z):
        return self._second_order(z)[2]

    def linearize_trajectory(self, X, U):
        A = np.zeros((len(X) - 1, self.nX, self.nX))
        B = np.zeros((len(X) - 1, self.nX, self.nU))
        for k in range(len(X) - 1):
            self.set(X[k], U[k], k, xk_hint=X[k + 1])
            A[k] = self.fdx()
            B[k] = self.fdu()
        return self.linearization_return(A, B)

    def second_order(self, Z):
        """(fdxdx, fdxdu, fdudu) for every trajectory: Z [B][nX] -> [B][nX][nX], [B][nX][nU], [B][nU][nU]."""
        hz = self.varint.deriv2_contract(Z)
        parts = [self._split_hz(hz[b]) for b in range(self.varint.batch)]
        return tuple(np.array([p[i] for p in parts]) for i in range(3))

    # -- trajectory files (dsystem.py:389-403) -------------------------------------------------------
    def save_state_trajectory(self, filename, X=None, U=None):
        from ..system import save_trajectory
        (Q, p, v, u, rho) = self.split_trajectory(X, U)
        save_trajectory(filename, self.system, self._time, Q, p, v, u, rho)

    def load_state_trajectory(self, filename):
        from ..system import load_trajectory
        (t, Q, p, v, u, rho) = load_trajectory(filename, self.system)
        self.time = t
        return self.build_trajectory(Q, p, v, u, rho)

    # -- projection onto the trajectory manifold (dsystem.py:426-494) ----------------------------------
    def project(self, bX, bU, Kproj=None):
        """X[0] = bX[0]; U[k] = bU[k] - Kproj[k] (X[k] - bX[k]); X[k+1] = f(X[k], U[k], k).
        The N dependent steps run as one closed-loop device rollout (one step size per step)."""
        bX, bU = np.asarray(bX, dtype=float), np.asarray(bU, dtype=float)
        if Kproj is None:
            Kproj = self.calc_feedback_controller(bX, bU)
        steps = np.diff(self._time)
        N = len(bX) - 1
        if N == 0:
            return self.trajectory_return(bX.copy(), bU.copy())
        eng = self._projection_engine()
        Q0, p0, _ = self.split_state(bX[0])
        eng.initialize_from_state(self._time[0], Q0[None], p0[None])
        nX, nU = eng.rollout_closed_loop(N, steps[:N], np.asarray(Kproj)[None], bX[None], bU[None])   # any time base
        _, status = eng.status()
        if status[0] != 0:
            from ..errors import ConvergenceError
            raise ConvergenceError("project: DEL solve failed")
        nX[0, 0, :] = bX[0]
        return self.trajectory_return(nX[0], nU[0])

    def _projection_engine(self):
        if getattr(self, "_proj", None) is None:
            self._proj = BatchMidpointVI(self.system, 1, tolerance=self.varint.tolerance, device=self.varint._device)
        self._proj.refresh()
        return self._proj

    def dproject(self, A, B, bdX, bdU, K):
        dX, dU = np.zeros(np.shape(bdX)), np.zeros(np.shape(bdU))
        dX[0] = bdX[0]
        for k in range(len(bdX) - 1):
            dU[k] = bdU[k] - np.dot(K[k], dX[k] - bdX[k])
            dX[k + 1] = np.dot(A[k], dX[k]) + np.dot(B[k], dU[k])
        return self.tangent_trajectory_return(dX, dU)

    def calc_feedback_controller(self, X, U, Q=None, R=None, return_linearization=False):
        from . import dlqr
        (A, B) = self.linearize_trajectory(X, U)
        if Q is None:
            eyeX = np.eye(self.nX)
            Q = lambda k: eyeX      # noqa: E731
        if R is None:
            eyeU = np.eye(self.nU)
            R = lambda k: eyeU      # noqa: E731
        Kproj = dlqr.solve_tv_lqr(A, B, Q, R)[0]
        return self.feedback_return(Kproj, A, B) if return_linearization else Kproj

    def convert_trajectory(self, dsys_a, X, U):
        """Map a trajectory of dsys_a onto this system by config / input names (dsystem.py:497-535)."""
        (qa, pa, va, ua, ra) = dsys_a.split_trajectory(X, U)
        qb, pb, vb = np.zeros((len(X), self._nQ)), np.zeros((len(X), self._np)), np.zeros((len(X), self._nv))
        ub, rb = np.zeros((len(U), self._nu)), np.zeros((len(U), self._nrho))

        def matches(list_a, list_b):
            names = [item.name for item in list_a]
            pairs = [(i, names.index(item.name)) for i, item in enumerate(list_b) if item.name in names]
            return [i for i, _ in pairs], [j for _, j in pairs]
        sa, sb = dsys_a.system, self.system
        for dst, src, (ib, ia) in ((qb, qa, matches(sa.configs, sb.configs)), (pb, pa, matches(sa.dyn_configs, sb.dyn_configs)),
                                   (ub, ua, matches(sa.inputs, sb.inputs)), (vb, va, matches(sa.kin_configs, sb.kin_configs)),
                                   (rb, ra, matches(sa.kin_configs, sb.kin_configs))):
            if ib:
                dst[:, ib] = src[:, ia]
        return self.build_trajectory(qb, pb, vb, ub, rb)

    # -- finite-difference validators (dsystem.py:538-704): every perturbation is one lane of a batch ------
    def _fd_batch(self, xk, uk, k, delta, wrt):
        n = self.nX if wrt == "x" else self.nU
        bd = BatchDSystem(self.system, self._time, 2 * n, device=self.varint._device, tolerance=self.varint.tolerance)
        Xp, Up = np.tile(np.asarray(xk, dtype=float), (2 * n, 1)), np.tile(np.asarray(uk, dtype=float), (2 * n, 1))
        tgt = Xp if wrt == "x" else Up
        tgt[np.arange(n), np.arange(n)] += delta
        tgt[n + np.arange(n), np.arange(n)] -= delta
        _, status = bd.set(Xp, Up, k)
        if (status != 0).any():
            from ..errors import ConvergenceError
            raise ConvergenceError("finite-difference check: DEL solve failed")
        return bd, n

    def _fd_report(self, exact, approx):
        return self.error_return(np.linalg.norm(exact - approx), np.linalg.norm(exact), np.linalg.norm(approx))

    def _check_first(self, xk, uk, k, delta, wrt):
        self.set(xk, uk, k)
        exact = self.fdx() if wrt == "x" else self.fdu()
        bd, n = self._fd_batch(xk, uk, k, delta, wrt)
        F = bd.f()
        bd.varint.close()
        return self._fd_report(exact, ((F[:n] - F[n:]) / (2 * delta)).T)

    def check_fdx(self, xk, uk, k, delta=1e-5):
        return self._check_first(xk, uk, k, delta, "x")

    def check_fdu(self, xk, uk, k, delta=1e-5):
        return self._check_first(xk, uk, k, delta, "u")

    def _check_second(self, xk, uk, k, delta, which):
        self.set(xk, uk, k)
        idx = {"xx": 0, "xu": 1, "uu": 2}[which]
        exact = np.array([self._second_order(z)[idx] for z in np.eye(self.nX)])       # [out][a][b]
        bd, n = self._fd_batch(xk, uk, k, delta, "u" if which[1] == "u" else "x")
        A, B = bd.linearize()
        bd.varint.close()
        J = A if which[0] == "x" else B                                                     # d f_out / d a
        approx = np.moveaxis((J[:n] - J[n:]) / (2 * delta), 0, 2)                           # [out][a][b]
        return self._fd_report(exact, approx)

    def check_fdxdx(self, xk, uk, k, delta=1e-5):
        return self._check_second(xk, uk, k, delta, "xx")

    def check_fdxdu(self, xk, uk, k, delta=1e-5):
        return self._check_second(xk, uk, k, delta, "xu")

    def check_fdudu(self, xk, uk, k, delta=1e-5):
        return self._check_second(xk, uk, k, delta, "uu")


class BatchDSystem(_Packing):
    """B independent copies of the discrete system on one GPU."""

    def __init__(self, system, t, batch, device=0, tolerance=1e-10, specialize="auto"):
        self.varint = BatchMidpointVI(system, batch, tolerance=tolerance, device=device, specialize=specialize)
        self._setup_packing(system, t)
        self._k = None

    system = property(lambda self: self.varint.system)
    batch = property(lambda self: self.varint.batch)
    k = property(lambda self: self._k)

    def set(self, Xk, Uk, k, Xk_hint=None, lambda_hint=None):
        """Xk [B][nX], Uk [B][nU]; returns (iterations[B], status[B])."""
        self._k = k
        Q, p, _ = self.split_state(Xk)
        u, rho = self.split_input(Uk)
        t1, t2 = self._time[k], self._time[k + 1]
        hint = None if Xk_hint is None else self.split_state(Xk_hint)[0][:, :self.varint.nd]
        self.varint.initialize_from_state(t1, Q, p)
        return self.varint.step(t2, u if self._nu else None, rho if self._nrho else None,
                                q2_hint=hint, lambda1_hint=lambda_hint)

    def f(self):
        v = self.varint
        q1, q2 = v.q1, v.q2
        t1, t2 = v.times()
        X = np.zeros((v.batch, self._nX))
        X[:, self._slice_Q] = q2
        X[:, self._slice_p] = v.p2
        X[:, self._slice_v] = (q2 - q1)[:, self._np:] / (t2 - t1)
        return X

    def linearize(self):
        """A [B][nX][nX], B [B][nX][nU] at the current step (one deriv1 launch for the batch)."""
        v = self.varint
        v.calc_deriv1()
        d = dict((n, np.swapaxes(v.deriv1(n), 1, 2)) for n in v.D1_NAMES[:8])  # -> output-major
        dt = self._time[self._k + 1] - self._time[self._k]
        A = np.zeros((v.batch, self._nX, self._nX))
        B = np.zeros((v.batch, self._nX, self._nU))
        for b in range(v.batch):
            A[b], B[b] = self._assemble(dt, d["q2_dq1"][b], d["q2_dp1"][b], d["q2_du1"][b], d["q2_dk2"][b],
                                        d["p2_dq1"][b], d["p2_dp1"][b], d["p2_du1"][b], d["p2_dk2"][b])
        return self.linearization_return(A, B)

    def second_order(self, Z):
        """(fdxdx, fdxdu, fdudu) for every trajectory: Z [B][nX] -> [B][nX][nX], [B][nX][nU], [B][nU][nU]."""
        hz = self.varint.deriv2_contract(Z)
        parts = [self._split_hz(hz[b]) for b in range(self.varint.batch)]
        return tuple(np.array([p[i] for p in parts]) for i in range(3))
