"""Quadratic tracking cost of a discrete trajectory (mirror of trep.discopt.DCost,
/root/reference/trep/discopt/dcost.py:5-118):
    l(x,u,k) = 1/2 (x-xd[k])' Q (x-xd[k]) + 1/2 (u-ud[k])' R (u-ud[k]),   m(xf) = 1/2 (xf-xd[-1])' Qf (xf-xd[-1]).
Besides the per-step accessors of the reference it offers whole-trajectory (vectorised) forms used by
the batched optimizer."""
import numpy as np


class DCost(object):
    def __init__(self, xd, ud, Q, R, Qf=None):
        self.xd = np.array(xd, dtype=float)
        self.ud = np.array(ud, dtype=float)
        self.Q = Q
        self.Qf = Q if Qf is None else Qf
        self.R = R
        self._S = np.zeros((Q.shape[0], R.shape[0]))

    # -- per step (reference API) ---------------------------------------------------------------
    def l(self, xk, uk, k):
        dx, du = xk - self.xd[k], uk - self.ud[k]
        return 0.5 * (dx.dot(self.Q).dot(dx) + du.dot(self.R).dot(du))

    def m(self, xkf):
        dx = xkf - self.xd[-1]
        return 0.5 * dx.dot(self.Qf).dot(dx)

    def l_dx(self, xk, uk, k):
        return (xk - self.xd[k]).dot(self.Q)

    def l_du(self, xk, uk, k):
        return (uk - self.ud[k]).dot(self.R)

    def m_dx(self, xkf):
        return (xkf - self.xd[-1]).dot(self.Qf)

    def l_dxdx(self, xk, uk, k):
        return self.Q.copy()

    def l_dxdu(self, xk, uk, k):
        return self._S.copy()

    def l_dudu(self, xk, uk, k):
        return self.R.copy()

    def m_dxdx(self, xkf):
        return self.Qf.copy()

    # -- whole trajectory -------------------------------------------------------------------------
    def total(self, X, U):
        """Cost of trajectories X [..., N+1, nX], U [..., N, nU] (leading batch axes allowed)."""
        dX = X[..., :-1, :] - self.xd[:-1]
        dU = U - self.ud
        dXf = X[..., -1, :] - self.xd[-1]
        run = 0.5 * (np.einsum("...ki,ij,...kj->...", dX, self.Q, dX) + np.einsum("...ki,ij,...kj->...", dU, self.R, dU))
        return run + 0.5 * np.einsum("...i,ij,...j->...", dXf, self.Qf, dXf)

    def gradients(self, X, U):
        """q [N+1][nX] (last row = terminal gradient) and r [N][nU]."""
        q = np.empty_like(X)
        q[:-1] = (X[:-1] - self.xd[:-1]).dot(self.Q)
        q[-1] = (X[-1] - self.xd[-1]).dot(self.Qf)
        r = (U - self.ud).dot(self.R)
        return q, r
