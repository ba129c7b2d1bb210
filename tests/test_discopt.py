"""discopt on top of the HIP integrator vs a DOptimizer trace recorded from the reference
(tests/golden/discopt_pend_on_cart.npz, made by tools/gen_golden.py::gen_discopt_cart)."""
import numpy as np
import pytest

from common import golden, relerr


def _problem():
    import trep_amd
    from trep_amd import systems, discopt
    g = golden("discopt_pend_on_cart")
    system = systems.pend_on_cart(torque_force=True)
    return g, system, discopt.DCost(g["Xd"], g["Ud"], g["Q"], g["R"])


def test_cost_functions_cpu():
    g, system, cost = _problem()
    X, U = g["X0"], g["U0"]
    assert abs(cost.total(X, U) - g["cost_initial"][0]) < 1e-9 * max(1.0, abs(g["cost_initial"][0]))
    per_step = sum(cost.l(X[k], U[k], k) for k in range(len(U))) + cost.m(X[-1])
    assert abs(per_step - cost.total(X, U)) < 1e-9 * max(1.0, per_step)
    q, r = cost.gradients(X, U)
    assert np.allclose(q[3], cost.l_dx(X[3], U[3], 3)) and np.allclose(q[-1], cost.m_dx(X[-1]))
    assert np.allclose(r[5], cost.l_du(X[5], U[5], 5))
    both = cost.total(np.stack([X, X]), np.stack([U, U]))
    assert both.shape == (2,) and abs(both[0] - both[1]) == 0.0


def test_tv_lqr_small_cpu():
    from trep_amd.discopt import dlqr
    rng = np.random.default_rng(0)
    N, nx, nu = 12, 3, 2
    A = rng.standard_normal((N, nx, nx)) * 0.3 + np.eye(nx)
    B = rng.standard_normal((N, nx, nu))
    Q, R = np.eye(nx), np.eye(nu)
    K, P = dlqr.solve_tv_lqr(A, B, lambda k: Q, lambda k: R)
    # one step of the Riccati recursion checked by hand at the last stage
    g = R + B[-1].T.dot(Q).dot(B[-1])
    assert np.allclose(K[-1], np.linalg.solve(g, B[-1].T.dot(Q).dot(A[-1])))
    assert np.allclose(P, P.T)
    q = rng.standard_normal((N + 1, nx)); r = rng.standard_normal((N, nu))
    K2, C, P2, b = dlqr.solve_tv_lq(A, B, q, r, lambda k: Q, lambda k: np.zeros((nx, nu)), lambda k: R)
    assert np.allclose(np.array(K2), np.array(K)) and np.allclose(P2, P)   # same quadratic part


@pytest.mark.gpu
def test_descent_directions_match_reference():
    import trep_amd
    from trep_amd import discopt
    g, system, cost = _problem()
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])
    opt = discopt.DOptimizer(dsys, cost)
    X, U = g["X0"], g["U0"]
    d = opt.calc_descent_direction(X, U, 'newton')
    assert relerr(np.array(d.Kproj), g["dd_Kproj"]) < 1e-7
    n = len(X)
    assert relerr(np.array([d.Q(k) for k in range(n)]), g["dd_newton_Q"]) < 1e-7
    assert relerr(np.array([d.S(k) for k in range(n - 1)]), g["dd_newton_S"]) < 1e-7
    assert relerr(np.array([d.R(k) for k in range(n - 1)]), g["dd_newton_R"]) < 1e-7
    assert relerr(d.dX, g["dd_newton_dX"]) < 1e-6 and relerr(d.dU, g["dd_newton_dU"]) < 1e-6
    d = opt.calc_descent_direction(X, U, 'quasi')
    assert relerr(d.dX, g["dd_quasi_dX"]) < 1e-6 and relerr(d.dU, g["dd_quasi_dU"]) < 1e-6


@pytest.mark.gpu
def test_optimizer_steps_match_reference_trace():
    import trep_amd
    from trep_amd import discopt
    g, system, cost = _problem()
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])

    class Rec(discopt.DOptimizerMonitor):
        def __init__(self):
            self.m = []

        def armijo_evaluation(self, armijo_iteration, nX, nU, bX, bU, cost, max_cost):
            self.m.append(armijo_iteration)

    mon = Rec()
    opt = discopt.DOptimizer(dsys, cost, monitor=mon)
    X, U = g["X0"].copy(), g["U0"].copy()
    for i, method in enumerate(g["methods"]):
        mon.m = []
        cost0 = opt.calc_cost(X, U)
        assert abs(cost0 - g["it%d_cost0" % i][0]) < 1e-7 * max(1.0, abs(cost0))
        (done, X, U, dcost0, cost1) = opt.step(i, X, U, str(method))
        assert not done
        assert abs(dcost0 - g["it%d_dcost0" % i][0]) < 1e-5 * abs(g["it%d_dcost0" % i][0])
        assert mon.m[-1] == int(g["it%d_m" % i][0])            # same accepted Armijo exponent
        assert abs(cost1 - g["it%d_cost1" % i][0]) < 1e-6 * max(1.0, abs(cost1))
        assert relerr(X, g["it%d_X" % i]) < 1e-5 and relerr(U, g["it%d_U" % i]) < 1e-5
    conv = opt.optimize(X, U, max_steps=2)
    assert conv.X.shape == X.shape


@pytest.mark.gpu
def test_closed_loop_rollout_equals_host_feedback_loop():
    """The in-kernel projection (U_k = bU_k - K_k (X_k - bX_k)) against the same loop driven step by step."""
    import trep_amd
    from trep_amd import systems, discopt
    g, system, cost = _problem()
    N = 60
    t = g["t"][:N + 1]
    rng = np.random.default_rng(1)
    bX, bU = g["X0"][:N + 1].copy(), g["U0"][:N].copy()
    bU += 0.3 * rng.standard_normal(bU.shape)
    K = 0.05 * rng.standard_normal((N, bU.shape[1], bX.shape[1]))
    eng = trep_amd.BatchMidpointVI(system, 2)
    dsb = discopt.BatchDSystem(system, t, 2)
    Q0, p0, _ = dsb.split_state(np.stack([bX[0], bX[0]]))
    eng.initialize_from_state(t[0], Q0, p0)
    nX, nU = eng.rollout_closed_loop(N, t[1] - t[0], K[None], np.stack([bX, bX]), np.stack([bU, bU]), group_size=2)
    one = discopt.DSystem(trep_amd.MidpointVI(system), t)
    x = bX[0].copy()
    for k in range(N):
        u = bU[k] - K[k].dot(x - bX[k])
        if k == 0:
            one.set(x, u, 0)
        else:
            one.step(u)
        x = one.f()
        assert relerr(nU[1, k], u) < 1e-9
        assert relerr(nX[0, k + 1], x) < 1e-9


@pytest.mark.gpu
def test_descent_model_validators_and_monitors(capsys):
    """check_dcost / check_ddcost (doptimizer.py:621-674), armijo_simulate and the monitor helpers on the pend-on-cart
    problem: the first and second directional derivatives of the cost agree with finite differences of projections."""
    import trep_amd
    from trep_amd import discopt
    g, system, cost = _problem()
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])
    opt = discopt.DOptimizer(dsys, cost, monitor=discopt.DOptimizerVerboseMonitor())
    X, U = g["X0"].copy(), g["U0"].copy()
    for method in ("quasi", "newton"):
        r = opt.check_dcost(X, U, method, delta=1e-5, tolerance=1e-4 * abs(opt.calc_dcost(X, U, *opt.calc_descent_direction(X, U, method)[1:3])))
        assert r.result, (method, r)
    r = opt.check_ddcost(X, U, "newton", delta=1e-5, tolerance=1e-3 * abs(g["it0_dcost0"][0]))
    assert abs(r.error) <= 1e-3 * abs(r.exact_ddcost), r
    Kproj = opt.calc_descent_direction(X, U, "quasi").Kproj
    ok = opt.armijo_simulate(X, U, Kproj)
    assert ok.success and relerr(ok.nX, X) < 1e-9
    (done, X1, U1, dcost0, cost1) = opt.step(0, X, U, "quasi")
    assert opt.monitor.get_costs()[-1] > cost1 and opt.monitor.get_dcosts()[-1] == dcost0
    assert "Armijo evaluation" in capsys.readouterr().out


@pytest.mark.gpu
def test_descent_curves_are_consistent_with_the_descent_model():
    """descent_curves (the data behind the reference's descent_plot, doptimizer.py:569-617): the projected cost along the
    direction is tangent to the quadratic model at z = 0, the accepted Armijo step lies below the required-improvement line."""
    import trep_amd
    from trep_amd import discopt
    g, system, cost = _problem()
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])
    opt = discopt.DOptimizer(dsys, cost)
    method = str(g["methods"][0])
    d = opt.descent_curves(g["X0"].copy(), g["U0"].copy(), method, points=12)
    assert len(d["z"]) == 32 and np.all(np.diff(d["z"]) >= 0) and len(d["armijo"]) == 20
    assert abs(d["dcost"] - g["it0_dcost0"][0]) < 1e-5 * abs(g["it0_dcost0"][0])
    small = (d["z"] > 0) & (d["z"] < 5e-3)
    assert small.sum() >= 3
    assert np.all(np.abs(d["true"][small] - d["model"][small]) <= 2e-2 * np.abs(d["model"][small]) + 1e-9)
    ok = d["armijo"] < d["required"][np.searchsorted(d["z"], d["armijo_z"])]
    assert ok.any()
    assert abs(d["armijo_z"][ok].max() - opt.armijo_beta ** int(g["it0_m"][0])) < 1e-15    # the step the reference trace accepted
    with pytest.raises(RuntimeError):
        import importlib.util
        if importlib.util.find_spec("matplotlib") is not None:
            raise RuntimeError("matplotlib present: plotting not exercised here")
        opt.descent_plot(g["X0"], g["U0"], "quasi", points=4)


# ---- non-uniform time base (the reference's DSystem takes any time vector, dsystem.py:229-274) -----------------------------
def _nonuniform():
    import trep_amd
    from trep_amd import systems, discopt
    g = golden("discopt_cart_nonuniform")
    system = systems.pend_on_cart(torque_force=True)
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])
    return g, system, dsys, discopt


@pytest.mark.gpu
def test_non_uniform_time_base_per_seed_optimizer_matches_reference():
    g, system, dsys, discopt = _nonuniform()
    assert np.ptp(np.diff(g["t"])) > 0.004                      # the time base really is non-uniform
    opt = discopt.DOptimizer(dsys, discopt.DCost(g["Xd"], g["Ud"], g["Q"], g["R"]))
    X, U = g["X0"].copy(), g["U0"].copy()
    A, B = opt.linearize(X, U)                                 # k-parallel: trajectory k steps by t[k+1] - t[k]
    assert relerr(A, g["lin_A"]) < 1e-8 and relerr(B, g["lin_B"]) < 1e-8
    lt = dsys.linearize_trajectory(X, U)                       # the sequential DSystem path
    assert relerr(lt.A, g["lin_A"]) < 1e-8 and relerr(lt.B, g["lin_B"]) < 1e-8
    d = opt.calc_descent_direction(X, U, 'newton')
    assert relerr(np.array(d.Kproj), g["dd_Kproj"]) < 1e-7
    assert relerr(d.dX, g["dd_newton_dX"]) < 1e-6 and relerr(d.dU, g["dd_newton_dU"]) < 1e-6
    d = opt.calc_descent_direction(X, U, 'quasi')
    assert relerr(d.dX, g["dd_quasi_dX"]) < 1e-6 and relerr(d.dU, g["dd_quasi_dU"]) < 1e-6
    for i, method in enumerate(g["methods"]):
        (done, X, U, dcost0, cost1) = opt.step(i, X, U, str(method))
        assert abs(dcost0 - g["it%d_dcost0" % i][0]) < 1e-5 * abs(g["it%d_dcost0" % i][0])
        assert abs(cost1 - g["it%d_cost1" % i][0]) < 1e-6 * max(1.0, abs(cost1))
        assert relerr(X, g["it%d_X" % i]) < 1e-5 and relerr(U, g["it%d_U" % i]) < 1e-5
    # DSystem.project on the non-uniform grid is one closed-loop device rollout with a step size per step
    pr = dsys.project(g["it0_X"], g["it0_U"], Kproj=np.array(d.Kproj))
    assert relerr(pr.X, g["it0_X"]) < 1e-7


@pytest.mark.gpu
def test_non_uniform_time_base_batch_optimizer_matches_reference():
    g, system, dsys, discopt = _nonuniform()
    S = 3
    rep = lambda a: np.repeat(a[None], S, axis=0)
    opt = discopt.BatchDOptimizer(dsys, rep(g["Xd"]), rep(g["Ud"]), g["Q"], g["R"])
    try:
        opt.set_trajectories(rep(g["X0"]), rep(g["U0"]))
        opt.linearize()
        assert relerr(opt.A.get()[1], g["lin_A"]) < 1e-8 and relerr(opt.B.get()[1], g["lin_B"]) < 1e-8
        for i, method in enumerate(g["methods"]):
            r = opt.step(str(method))
            X, U = opt.get_trajectories()
            for s in range(S):
                assert abs(r.dcost0[s] - g["it%d_dcost0" % i][0]) < 1e-5 * abs(g["it%d_dcost0" % i][0])
                assert r.armijo[s] == int(g["it%d_m" % i][0])
                assert abs(r.cost1[s] - g["it%d_cost1" % i][0]) < 1e-6 * max(1.0, abs(r.cost1[s]))
                assert relerr(X[s], g["it%d_X" % i]) < 1e-5
    finally:
        opt.close()
