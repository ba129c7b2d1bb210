"""Time-varying discrete LQR / LQ backward sweeps (what trep.discopt.dlqr provides,
/root/reference/trep/discopt/dlqr.py:9-81), on stacked arrays A [N][nX][nX], B [N][nX][nU].
Cost weights are callables k -> matrix like in the reference.  Host-side numpy: the sweeps are
sequential in k and tiny next to the k-parallel linearisation that feeds them."""
from collections import namedtuple

import numpy as np

solve_tv_lqr_return = namedtuple('solve_tv_lqr', 'K P')
solve_tv_lq_return = namedtuple('solve_tv_lq', 'K C P b')


def solve_tv_lqr(A, B, Q, R):
    """Riccati recursion P_k = Q_k + A'PA - (B'PA)' (R_k + B'PB)^-1 (B'PA); returns gains K [N][nU][nX], P_0."""
    kf = len(A)
    K = [None] * kf
    P = Q(kf)
    for k in range(kf - 1, -1, -1):
        BtP = B[k].T.dot(P)
        gamma = R(k) + BtP.dot(B[k])
        K_part = BtP.dot(A[k])
        K[k] = np.linalg.solve(gamma, K_part)
        P = Q(k) + A[k].T.dot(P).dot(A[k]) - K_part.T.dot(K[k])
        P = (P + P.T) / 2.0      # keeps the recursion symmetric (needed for stability)
    return solve_tv_lqr_return(K, P)


def solve_tv_lq(A, B, q, r, Q, S, R):
    """Affine LQ problem with linear terms q [N+1], r [N] and cross weight S(k): gains K, offsets C, P_0, b_0."""
    kf = len(A)
    K = [None] * kf
    C = [None] * kf
    P = Q(kf)
    b = q[kf]
    for k in range(kf - 1, -1, -1):
        BtP = B[k].T.dot(P)
        gamma = R(k) + BtP.dot(B[k])
        K_part = BtP.dot(A[k]) + S(k).T
        sol = np.linalg.solve(gamma, np.column_stack([B[k].T.dot(b) + r[k], K_part]))
        C[k] = sol[:, 0]
        K[k] = sol[:, 1:]
        b = q[k] - K[k].T.dot(r[k]) + (A[k].T - K[k].T.dot(B[k].T)).dot(b)
        P = Q(k) + A[k].T.dot(P).dot(A[k]) - K_part.T.dot(K[k])
        P = (P + P.T) / 2.0
    return solve_tv_lq_return(K, C, P, b)
