#!/usr/bin/env python3
"""Golden vectors for the DSystem helpers around the hot path, from the REAL reference.

Build container only (needs /tmp/trep_ref from tools/build_reference.py).  Writes
  tests/golden/pend_on_cart_traj.mat   -- a trajectory file written by the reference's trep.save_trajectory
  tests/golden/dsystem_extras.npz      -- DSystem.project / dproject / calc_feedback_controller / convert_trajectory
                                           outputs of the reference for the pend-on-cart and scissor systems.
Data only.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/tmp/trep_ref")
if not hasattr(np, "object"):
    np.object = object   # the reference predates numpy 1.24

import trep  # noqa: E402
import trep.discopt  # noqa: E402
from trep_amd import systems  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
DT = 0.01


def main():
    rng = np.random.default_rng(77)
    out = {}
    system = systems.pend_on_cart(api=trep)
    N = 40
    t = DT * np.arange(N + 1)
    mvi = trep.MidpointVI(system)
    dsys = trep.discopt.DSystem(mvi, t)
    # a true trajectory under random inputs
    q0 = np.array([0.3, 2.0])
    mvi.initialize_from_configs(0.0, q0, DT, q0)
    X0 = dsys.build_state(Q=mvi.q2, p=mvi.p2)
    U = 2.0 * rng.standard_normal((N, dsys.nU))
    X = np.zeros((N + 1, dsys.nX))
    X[0] = X0
    dsys.set(X[0], U[0], 0)
    X[1] = dsys.f()
    for k in range(1, N):
        dsys.step(U[k])
        X[k + 1] = dsys.f()
    (Q, p, v, u, rho) = dsys.split_trajectory(X, U)
    trep.save_trajectory(os.path.join(OUT, "pend_on_cart_traj.mat"), system, t, Q, p, v, u, rho)
    out["poc_X"], out["poc_U"] = X, U
    Kproj = dsys.calc_feedback_controller(X, U)
    out["poc_Kproj"] = Kproj
    Qw = np.diag([3.0, 2.0, 1.0, 0.5])
    Rw = np.diag([0.7])
    out["poc_Qw"], out["poc_Rw"] = Qw, Rw
    (K2, A, B) = dsys.calc_feedback_controller(X, U, lambda k: Qw, lambda k: Rw, return_linearization=True)
    out["poc_K2"], out["poc_A"], out["poc_B"] = K2, A, B
    bX = X + 0.02 * rng.standard_normal(X.shape)
    bU = U + 0.1 * rng.standard_normal(U.shape)
    out["poc_bX"], out["poc_bU"] = bX, bU
    (pX, pU) = dsys.project(bX, bU, Kproj)
    out["poc_pX"], out["poc_pU"] = pX, pU
    bdX = rng.standard_normal(X.shape)
    bdU = rng.standard_normal(U.shape)
    out["poc_bdX"], out["poc_bdU"] = bdX, bdU
    (dX, dU) = dsys.dproject(A, B, bdX, bdU, Kproj)
    out["poc_dX"], out["poc_dU"] = dX, dU
    for name in ("check_fdx", "check_fdu", "check_fdxdx", "check_fdxdu", "check_fdudu"):
        out["poc_" + name] = np.array(getattr(dsys, name)(X[7], U[7], 7))

    # convert_trajectory: pendulum(3) -> pendulum(5) shares the first three link names
    sa, sb = systems.pendulum(3, api=trep), systems.pendulum(5, api=trep)
    da = trep.discopt.DSystem(trep.MidpointVI(sa), t[:6])
    db = trep.discopt.DSystem(trep.MidpointVI(sb), t[:6])
    Xa = rng.standard_normal((6, da.nX))
    Ua = rng.standard_normal((5, da.nU))
    (Xb, Ub) = db.convert_trajectory(da, Xa, Ua)
    out["conv_Xa"], out["conv_Ua"], out["conv_Xb"], out["conv_Ub"] = Xa, Ua, Xb, Ub
    # System.minimize_potential_energy (system.py:215-272) on the spring_link system, from its builder pose
    sl = systems.spring_link(api=trep)
    sl.q = {'a': 0.3, 'b': -0.2, 'd': 0.4, 'e': -1.0, 'slide': 0.1}
    out["minpot_q0"] = sl.q
    out["minpot_q"] = sl.minimize_potential_energy(keep_kinematic=True)
    out["minpot_V"] = np.array([-sl.L()])
    np.savez_compressed(os.path.join(OUT, "dsystem_extras.npz"), **out)
    print("wrote", sorted(out))


if __name__ == "__main__":
    main()
