"""DSystem helpers either side of the hot path: trajectory files, dproject,
convert_trajectory on the host; project / calc_feedback_controller / check_fd* through the HIP integrator.
Golden data: tools/gen_dsystem_fixture.py (the real reference)."""
import os
import types

import numpy as np
import pytest

import trep_amd
from trep_amd import systems, discopt
from common import GOLDEN, relerr

DT = 0.01


def extras():
    return dict(np.load(os.path.join(GOLDEN, "dsystem_extras.npz")))


def host_dsystem(system, t):
    """A DSystem whose integrator is never touched (packing / file / tangent helpers only)."""
    return discopt.DSystem(types.SimpleNamespace(system=system), t)


def test_load_trajectory_written_by_the_reference():
    g = extras()
    system = systems.pend_on_cart()
    path = os.path.join(GOLDEN, "pend_on_cart_traj.mat")
    ds = host_dsystem(system, [0.0, 1.0])
    X, U = ds.load_state_trajectory(path)
    assert np.array_equal(X, g["poc_X"]) and np.array_equal(U, g["poc_U"])
    assert np.allclose(ds.time, DT * np.arange(len(X)), rtol=0, atol=1e-15)
    t, (Qn, Q), (pn, p), (vn, v), (un, u), (rn, rho) = trep_amd.load_trajectory(path)
    assert Qn == [c.name for c in system.configs] and pn == [c.name for c in system.dyn_configs]
    assert un == [i.name for i in system.inputs] and vn == [] and rn == []
    assert np.array_equal(Q, g["poc_X"][:, :2]) and np.array_equal(u, g["poc_U"])


def test_save_trajectory_round_trip_and_renamed_system(tmp_path):
    system = systems.puppet()
    ds = host_dsystem(system, DT * np.arange(6))
    rng = np.random.default_rng(5)
    X, U = rng.standard_normal((6, ds.nX)), rng.standard_normal((5, ds.nU))
    path = str(tmp_path / "traj.mat")
    ds.save_state_trajectory(path, X, U)
    ds2 = host_dsystem(systems.puppet(), [0.0, 1.0])
    X2, U2 = ds2.load_state_trajectory(path)
    assert np.array_equal(X2, X) and np.array_equal(U2, U) and ds2.kf() == 5
    # a different system: only the columns with matching names are filled
    other = systems.puppet_basic()
    t, Q, p, v, u, rho = trep_amd.load_trajectory(path, other)
    src = dict((c.name, X[:, c.index]) for c in system.configs)
    for c in other.configs:
        expect = src.get(c.name, np.zeros(6))
        assert np.array_equal(Q[:, c.index], expect), c.name
    assert v.shape == (6, other.nQk) and rho.shape == (5, other.nQk)


def test_dproject_and_convert_trajectory_match_reference():
    g = extras()
    ds = host_dsystem(systems.pend_on_cart(), DT * np.arange(len(g["poc_X"])))
    dX, dU = ds.dproject(g["poc_A"], g["poc_B"], g["poc_bdX"], g["poc_bdU"], g["poc_Kproj"])
    assert relerr(dX, g["poc_dX"]) < 1e-13 and relerr(dU, g["poc_dU"]) < 1e-13
    t = DT * np.arange(6)
    da, db = host_dsystem(systems.pendulum(3), t), host_dsystem(systems.pendulum(5), t)
    Xb, Ub = db.convert_trajectory(da, g["conv_Xa"], g["conv_Ua"])
    assert np.array_equal(Xb, g["conv_Xb"]) and np.array_equal(Ub, g["conv_Ub"])


@pytest.mark.gpu
def test_project_and_feedback_controller_match_reference():
    g = extras()
    system = systems.pend_on_cart()
    X, U = g["poc_X"], g["poc_U"]
    ds = discopt.DSystem(trep_amd.MidpointVI(system), DT * np.arange(len(X)))
    K = ds.calc_feedback_controller(X, U)
    assert relerr(K, g["poc_Kproj"]) < 1e-9
    fb = ds.calc_feedback_controller(X, U, lambda k: g["poc_Qw"], lambda k: g["poc_Rw"], return_linearization=True)
    assert relerr(fb.Kproj, g["poc_K2"]) < 1e-9 and relerr(fb.A, g["poc_A"]) < 1e-10 and relerr(fb.B, g["poc_B"]) < 1e-10
    pX, pU = ds.project(g["poc_bX"], g["poc_bU"], g["poc_Kproj"])
    assert relerr(pX, g["poc_pX"]) < 1e-10 and relerr(pU, g["poc_pU"]) < 1e-10
    assert np.array_equal(pX[0], g["poc_bX"][0])
    # non-uniform time base: the step-by-step branch
    ds2 = discopt.DSystem(trep_amd.MidpointVI(system), DT * np.arange(len(X)))
    uniform = ds2.time.copy()
    ds2.time = np.concatenate([uniform[:-1], [uniform[-1] + 1e-10]])
    qX, qU = ds2.project(g["poc_bX"], g["poc_bU"], g["poc_Kproj"])
    assert relerr(qX, g["poc_pX"]) < 1e-7


@pytest.mark.gpu
def test_finite_difference_validators_match_reference():
    """check_fd* report (error, exact_norm, approx_norm) like the reference's (dsystem.py:538-704)."""
    g = extras()
    X, U = g["poc_X"], g["poc_U"]
    ds = discopt.DSystem(trep_amd.MidpointVI(systems.pend_on_cart()), DT * np.arange(len(X)))
    for name in ("check_fdx", "check_fdu", "check_fdxdx", "check_fdxdu", "check_fdudu"):
        err, exact, approx = getattr(ds, name)(X[7], U[7], 7)
        ref = g["poc_" + name]
        assert abs(exact - ref[1]) <= 1e-9 * max(1.0, ref[1]), name
        assert abs(approx - ref[2]) <= 1e-5 * max(1.0, ref[2]), name
        assert err <= max(10 * ref[0], 1e-6), (name, err, ref[0])


@pytest.mark.gpu
def test_minimize_potential_energy_matches_reference():
    """System.minimize_potential_energy (system.py:215-272) on the spring_link system: same equilibrium and energy as the
    reference's SLSQP run from the same pose (V and its finite-difference gradient come from the energy kernel)."""
    g = extras()
    system = systems.spring_link()
    system.q = g["minpot_q0"]
    q = system.minimize_potential_energy(keep_kinematic=True)
    assert abs(-system.L() - g["minpot_V"][0]) < 1e-7
    assert np.allclose(np.cos(q - g["minpot_q"]), 1.0, atol=1e-8) or relerr(q, g["minpot_q"]) < 1e-4
    assert max(abs(c.h()) for c in system.constraints) < 1e-8 and np.all(system.dq == 0)
