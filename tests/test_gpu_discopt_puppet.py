"""BASELINE config 4 (examples/puppet-optimization.py: discopt on the ~40-DOF puppet) pinned to the reference.

tests/golden/discopt_puppet.npz is a trace of the REFERENCE's DOptimizer on that problem at a short horizon
(tools/gen_golden.py::gen_discopt_puppet: N = 50, one quasi-Newton and one Newton step; projection gain, Newton model
Q/S/R, both descent directions, cost0 / dcost0 / accepted Armijo exponent / cost1 / new trajectory per step).  The
per-seed ``DOptimizer`` and the device-resident ``BatchDOptimizer`` are both compared with it; then the full-size
configuration (256 seeds x N = 1000) is run through size-independent properties.

Tolerances: cost 1e-9 relative, gains / model / directions 1e-6 relative to the array's largest entry (they pass
through N Riccati steps and second derivatives pinned at 1e-8), trajectories after a step 1e-6."""
import numpy as np
import pytest

from common import golden, relerr

pytestmark = pytest.mark.gpu


def _problem():
    import trep_amd
    from trep_amd import systems, discopt
    g = golden("discopt_puppet")
    system = systems.puppet()
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), g["t"])
    return g, system, dsys


class _Rec(object):
    def __init__(self):
        self.m = []

    def __getattr__(self, name):          # every other monitor hook: ignore
        return lambda *a, **k: None

    def armijo_evaluation(self, armijo_iteration, nX, nU, bX, bU, cost, max_cost):
        self.m.append(armijo_iteration)


def test_initial_trajectories_match_reference():
    """The desired and the initial trajectory of the problem (two 50-step puppet rollouts through DSystem)."""
    g, system, dsys = _problem()
    for X, U in ((g["Xd"], g["Ud"]), (g["X0"], g["U0"])):
        Y = np.zeros_like(X)
        Y[0] = X[0]
        for k in range(len(U)):
            if k == 0:
                dsys.set(X[0], U[0], 0)
            else:
                dsys.step(U[k])
            Y[k + 1] = dsys.f()
        assert relerr(Y, X) < 1e-10


def test_per_seed_descent_directions_match_reference():
    from trep_amd import discopt
    g, system, dsys = _problem()
    opt = discopt.DOptimizer(dsys, discopt.DCost(g["Xd"], g["Ud"], g["Q"], g["R"]))
    X, U = g["X0"], g["U0"]
    N = len(U)
    assert abs(opt.calc_cost(X, U) - g["cost_initial"][0]) < 1e-9 * g["cost_initial"][0]
    d = opt.calc_descent_direction(X, U, 'newton')
    assert relerr(np.array(d.Kproj), g["dd_Kproj"]) < 1e-6
    ks = g["model_k"]
    assert relerr(np.array([d.Q(k) for k in ks]), g["dd_newton_Q"]) < 1e-6
    assert relerr(d.Q(N), g["dd_newton_Qf"]) < 1e-12
    assert relerr(np.array([d.S(k) for k in ks]), g["dd_newton_S"]) < 1e-6
    assert relerr(np.array([d.R(k) for k in ks]), g["dd_newton_R"]) < 1e-6
    assert relerr(d.dX, g["dd_newton_dX"]) < 1e-6 and relerr(d.dU, g["dd_newton_dU"]) < 1e-6
    assert abs(opt.calc_dcost(X, U, d.dX, d.dU) - g["dd_newton_dcost"][0]) < 1e-6 * abs(g["dd_newton_dcost"][0])
    d = opt.calc_descent_direction(X, U, 'quasi')
    assert relerr(d.dX, g["dd_quasi_dX"]) < 1e-6 and relerr(d.dU, g["dd_quasi_dU"]) < 1e-6
    assert abs(opt.calc_dcost(X, U, d.dX, d.dU) - g["dd_quasi_dcost"][0]) < 1e-6 * abs(g["dd_quasi_dcost"][0])


def test_per_seed_steps_match_reference_trace():
    from trep_amd import discopt
    g, system, dsys = _problem()
    mon = _Rec()
    opt = discopt.DOptimizer(dsys, discopt.DCost(g["Xd"], g["Ud"], g["Q"], g["R"]), monitor=mon)
    X, U = g["X0"].copy(), g["U0"].copy()
    for i, method in enumerate(g["methods"]):
        mon.m = []
        cost0 = opt.calc_cost(X, U)
        assert abs(cost0 - g["it%d_cost0" % i][0]) < 1e-8 * max(1.0, abs(cost0))
        (done, X, U, dcost0, cost1) = opt.step(i, X, U, str(method))
        assert not done
        assert abs(dcost0 - g["it%d_dcost0" % i][0]) < 1e-6 * abs(g["it%d_dcost0" % i][0])
        assert mon.m[-1] == int(g["it%d_m" % i][0])            # the reference's accepted Armijo exponent
        assert abs(cost1 - g["it%d_cost1" % i][0]) < 1e-7 * max(1.0, abs(cost1))
        assert relerr(X, g["it%d_X" % i]) < 1e-6 and relerr(U, g["it%d_U" % i]) < 1e-6


def test_batch_optimizer_matches_reference_trace():
    """Three seeds in the device-resident optimiser: the reference's problem twice (seeds 0 and 2) around a different
    one (seed 1, another initial pose) -- seeds 0 and 2 must reproduce the reference trace and each other bit for bit."""
    from trep_amd import discopt, systems
    import trep_amd
    g, system, dsys = _problem()
    N, dt, nd = len(g["U0"]), float(g["t"][1] - g["t"][0]), system.nQd
    Q1 = systems.puppet_initial_conditions(system, 1, seed=5)
    K_move = systems.puppet_string_schedule(system, Q1[:, nd:], N, dt)
    K_still = np.repeat(Q1[:, None, nd:], N, axis=1)
    sim = trep_amd.BatchMidpointVI(system, 1)
    sim.initialize_from_state(0.0, Q1, np.zeros((1, nd)))
    Xd1 = sim.rollout(N, dt, None, K_move)[0]
    sim.initialize_from_state(0.0, Q1, np.zeros((1, nd)))
    Xi1 = sim.rollout(N, dt, None, K_still)[0]
    sim.close()
    Xd = np.stack([g["Xd"], Xd1, g["Xd"]]); Ud = np.stack([g["Ud"], K_move[0], g["Ud"]])
    Xi = np.stack([g["X0"], Xi1, g["X0"]]); Ui = np.stack([g["U0"], K_still[0], g["U0"]])
    opt = discopt.BatchDOptimizer(dsys, Xd, Ud, g["Q"], g["R"])
    try:
        opt.set_trajectories(Xi, Ui)
        # the stages of one Newton step, compared one by one with the reference's calc_descent_direction
        opt.linearize()
        opt.projection_gain()
        cost0 = opt.gradients_and_cost()
        assert abs(cost0[0] - g["cost_initial"][0]) < 1e-9 * g["cost_initial"][0] and cost0[0] == cost0[2]
        Kproj = opt.Kproj.get()
        assert relerr(Kproj[0], g["dd_Kproj"]) < 1e-6 and np.array_equal(Kproj[0], Kproj[2])
        opt.descent_direction(None, "newton")
        HZ = opt.HZ.get()
        for j, k in enumerate(g["model_k"]):
            xx, xu, uu = dsys._split_hz(HZ[0][k])
            assert relerr(g["Q"] + xx, g["dd_newton_Q"][j]) < 1e-6
            assert relerr(xu, g["dd_newton_S"][j]) < 1e-6
            assert relerr(g["R"] + uu, g["dd_newton_R"][j]) < 1e-6
        dX, dU, dc = opt.dX.get(), opt.dU.get(), opt.dcost.get()
        assert relerr(dX[0], g["dd_newton_dX"]) < 1e-6 and relerr(dU[0], g["dd_newton_dU"]) < 1e-6
        assert abs(dc[0] - g["dd_newton_dcost"][0]) < 1e-6 * abs(g["dd_newton_dcost"][0])
        opt.descent_direction(None, "quasi")
        dX, dU, dc = opt.dX.get(), opt.dU.get(), opt.dcost.get()
        assert relerr(dX[0], g["dd_quasi_dX"]) < 1e-6 and relerr(dU[0], g["dd_quasi_dU"]) < 1e-6
        assert abs(dc[0] - g["dd_quasi_dcost"][0]) < 1e-6 * abs(g["dd_quasi_dcost"][0])
        # whole steps
        for i, method in enumerate(g["methods"]):
            r = opt.step(str(method))
            X, U = opt.get_trajectories()
            assert not r.failed.any() and not r.done.any()
            for s in (0, 2):
                assert abs(r.cost0[s] - g["it%d_cost0" % i][0]) < 1e-8 * max(1.0, abs(r.cost0[s]))
                assert abs(r.dcost0[s] - g["it%d_dcost0" % i][0]) < 1e-6 * abs(g["it%d_dcost0" % i][0])
                assert r.armijo[s] == int(g["it%d_m" % i][0])
                assert abs(r.cost1[s] - g["it%d_cost1" % i][0]) < 1e-7 * max(1.0, abs(r.cost1[s]))
                assert relerr(X[s], g["it%d_X" % i]) < 1e-6 and relerr(U[s], g["it%d_U" % i]) < 1e-6
            assert np.array_equal(X[0], X[2]) and np.array_equal(U[0], U[2])
            assert r.cost1[1] < r.cost0[1]
    finally:
        opt.close()


# ---- BASELINE config 4 at full size: 256 seeds x N = 1000 ----------------------------------------------------------

def _full_problem(S, N):
    """bench_discopt.problem: S perturbed puppet poses, desired = moving strings, initial guess = still strings."""
    import bench_discopt
    return bench_discopt.problem(S, N, 0.01)


def test_full_size_256_seeds_1000_steps_properties():
    """One quasi-Newton and one Newton step of all 256 seeds at N = 1000.  Size-independent properties:
      * every accepted step lowers that seed's cost and satisfies the Armijo inequality;
      * duplicated seeds (the first 8 problems are repeated as the last 8) are bit-identical;
      * the first 32 seeds run as their own batch (what one of 8 GPUs holds) are bit-identical to the same seeds
        inside the 256 batch;
      * the result is a trajectory: X[s] is reproduced by an open-loop rollout of U[s] from X[s][0];
      * every seed flagged `failed` (Armijo exhausted) also fails in the per-seed DOptimizer, from the same iterate."""
    import trep_amd
    from trep_amd import discopt
    from trep_amd.errors import ConvergenceError
    S, N, dt = 256, 1000, 0.01
    system, Xd, Ud, Xi, Ui, Qc, Rc = _full_problem(S - 8, N)
    dup = lambda a: np.concatenate([a, a[:8]], axis=0)
    Xd, Ud, Xi, Ui = dup(Xd), dup(Ud), dup(Xi), dup(Ui)
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), dt * np.arange(N + 1))
    methods = ["quasi", "newton"]
    opt = discopt.BatchDOptimizer(dsys, Xd, Ud, Qc, Rc)
    results, iterates = [], []
    try:
        opt.set_trajectories(Xi, Ui)
        for m in methods:
            iterates.append(opt.get_trajectories())
            results.append(opt.step(m))
        X, U = opt.get_trajectories()
    finally:
        opt.close()
    for r in results:
        okay = ~r.failed & ~r.done
        assert okay.sum() >= S - 16, "too many failed seeds: %d" % r.failed.sum()
        assert (r.cost1[okay] < r.cost0[okay]).all()
        lam = 0.7 ** r.armijo[okay]
        assert (r.cost1[okay] < r.cost0[okay] + 1e-5 * lam * r.dcost0[okay]).all()
        assert (r.dcost0[okay] < 0).all()
        assert np.array_equal(r.cost1[:8], r.cost1[-8:]) and np.array_equal(r.armijo[:8], r.armijo[-8:])
    assert np.array_equal(X[:8], X[-8:]) and np.array_equal(U[:8], U[-8:])
    assert np.isfinite(X).all() and np.isfinite(U).all()
    # the 32-seed shard
    sub = discopt.BatchDOptimizer(dsys, Xd[:32], Ud[:32], Qc, Rc, armijo_chunk=opt.M)
    try:
        sub.set_trajectories(Xi[:32], Ui[:32])
        for m, r in zip(methods, results):
            rs = sub.step(m)
            assert np.array_equal(rs.cost1, r.cost1[:32], equal_nan=True) and np.array_equal(rs.armijo, r.armijo[:32])
        Xs, Us = sub.get_trajectories()
    finally:
        sub.close()
    assert np.array_equal(Xs, X[:32]) and np.array_equal(Us, U[:32])
    # the optimised (X, U) are trajectories of the system
    nd = system.nQd
    sim = trep_amd.BatchMidpointVI(system, S)
    sim.initialize_from_state(0.0, X[:, 0, :system.nQ], X[:, 0, system.nQ:system.nQ + nd])
    Xr = sim.rollout(N, dt, None, U)
    sim.close()
    assert relerr(Xr[:, :, :system.nQ + nd], X[:, :, :system.nQ + nd]) < 1e-7
    # failed seeds fail per seed too (the reference raises ConvergenceError("Armijo Failed to Converge") there)
    checked = 0
    for it, r in enumerate(results):
        for s in np.nonzero(r.failed)[0][:3]:
            o = discopt.DOptimizer(dsys, discopt.DCost(Xd[s], Ud[s], Qc, Rc))
            Xs0, Us0 = iterates[it][0][s], iterates[it][1][s]
            with pytest.raises(ConvergenceError):
                o.step(it, Xs0, Us0, methods[it])
            checked += 1
    print("full-size discopt: failed per step %s, per-seed cross-checks %d" % ([int(r.failed.sum()) for r in results], checked))
