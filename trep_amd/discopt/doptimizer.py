"""DOptimizer: projection-operator trajectory optimisation on top of the batched integrator.

Same algorithm and public surface as the reference's ``trep.discopt.DOptimizer``
(/root/reference/trep/discopt/doptimizer.py:207-566): per iteration a descent direction from a
time-varying LQ problem (steepest / quasi-Newton / Newton model), an Armijo search along it with
every candidate projected back onto the trajectory manifold by a closed-loop rollout, fallback
newton -> quasi -> steepest, termination on |dcost| < descent_tolerance.

What is different is *where the parallelism is* (the reference is a serial Python loop over k):

* ``linearize``: the N teacher-forced DEL solves ``set(X[k], U[k], k, xk_hint=X[k+1])`` of
  ``DSystem.linearize_trajectory`` (dsystem.py:406-423) are independent in k, so all of them run as ONE
  batch of N trajectories (one step launch + one deriv1 launch) -> A [N][nX][nX], B [N][nX][nU];
* Newton model (doptimizer.py:319-345): the adjoint recursion z_k only needs A, B and the projection gain,
  so it is done on the host first; the N second-derivative contractions fdxdx(z_k), fdxdu(z_k),
  fdudu(z_k) then run as one batched deriv2 launch on the same N solved steps;
* Armijo (doptimizer.py:405-459): all candidates lambda = beta^m, m < armijo_max_iterations, are rolled
  out together (closed-loop, feedback evaluated in the kernel); the first m that passes the sufficient
  decrease test is accepted, which is exactly what the sequential search returns.

The Riccati / LQ sweeps (dlqr.py) stay on the host: sequential in k, O(N nX^3), negligible.
Any time base works: the horizon batch steps trajectory k by t[k+1] - t[k], the rollouts take one step size per step.
"""
from collections import namedtuple

import numpy as np

from . import dlqr
from ..errors import ConvergenceError
from ..midpointvi import BatchMidpointVI


class DOptimizerMonitor(object):
    """Callback sink (doptimizer.py:19-112); every hook is optional."""

    def optimize_begin(self, X, U): pass
    def optimize_end(self, converged, X, U, cost): pass
    def step_begin(self, iteration): pass
    def step_info(self, method, cost, dcost, X, U, dX, dU, Kproj): pass
    def step_method_failure(self, method, cost, dcost, fallback_method): pass
    def step_termination(self, cost, dcost): pass
    def step_completed(self, method, cost, nX, nU): pass
    def armijo_simulation_failure(self, armijo_iteration, nX, nU, bX, bU): pass
    def armijo_search_failure(self, X, U, dX, dU, cost0, dcost0, Kproj): pass
    def armijo_evaluation(self, armijo_iteration, nX, nU, bX, bU, cost, max_cost): pass


class DOptimizerDefaultMonitor(DOptimizerMonitor):
    """Keeps cost / dcost histories like the reference's default monitor (doptimizer.py:115-205), silently."""

    def __init__(self):
        self.iteration = 0
        self.cost_history = {}
        self.dcost_history = {}

    def step_begin(self, iteration):
        self.iteration = iteration

    def step_info(self, method, cost, dcost, X, U, dX, dU, Kproj):
        self.cost_history[self.iteration] = cost
        self.dcost_history[self.iteration] = dcost

    def get_costs(self):
        return [self.cost_history[i] for i in sorted(self.cost_history)]

    def get_dcosts(self):
        return [self.dcost_history[i] for i in sorted(self.dcost_history)]

    def msg(self, text):
        import datetime
        print("%s %3d: %s" % (datetime.datetime.now().strftime('[%H:%M:%S]'), self.iteration, text))


class DOptimizerVerboseMonitor(DOptimizerDefaultMonitor):
    """Prints what the optimiser does (doptimizer.py:180-246)."""

    def optimize_begin(self, X, U):
        self.msg("Optimization starting")

    def optimize_end(self, converged, X, U, cost):
        self.msg("Optimization completed%s, cost %f" % ("" if converged else " (not converged)", cost))

    def step_info(self, method, cost, dcost, X, U, dX, dU, Kproj):
        self.msg("Current Trajectory cost: %f, dcost: %f, method=%s" % (cost, dcost, method))
        DOptimizerDefaultMonitor.step_info(self, method, cost, dcost, X, U, dX, dU, Kproj)

    def step_method_failure(self, method, cost, dcost, fallback_method):
        self.msg("Descent method %r failed (dcost %f), falling back to %r" % (method, dcost, fallback_method))

    def step_termination(self, cost, dcost):
        self.msg("Optimization terminated: cost %f, dcost %f" % (cost, dcost))

    def armijo_evaluation(self, armijo_iteration, nX, nU, bX, bU, cost, max_cost):
        verdict = "is too expensive" if cost >= max_cost else "is acceptable"
        self.msg("  Armijo evaluation (%d) %s (%f vs %f)" % (armijo_iteration, verdict, cost, max_cost))

    def armijo_simulation_failure(self, armijo_iteration, nX, nU, bX, bU):
        self.msg("  Armijo simulation (%d) failed" % armijo_iteration)


class DOptimizer(object):
    def __init__(self, dsys, cost, first_method_iterations=10, monitor=None, device=0):
        self.dsys = dsys
        self.cost = cost
        self.optimize_ic = False
        self.monitor = DOptimizerDefaultMonitor() if monitor is None else monitor
        Qproj = np.eye(self.dsys.nX)
        Rproj = np.eye(self.dsys.nU)
        self.Qproj = lambda k: Qproj
        self.Rproj = lambda k: Rproj
        self.armijo_beta = 0.7
        self.armijo_alpha = 0.00001
        self.armijo_max_iterations = 30
        self.descent_tolerance = 1e-6
        self.first_method_iterations = first_method_iterations
        self.first_method = 'quasi'
        self.second_method = 'newton'
        self.step_return = namedtuple('step', 'done nX nU dcost0 cost1')
        self.optimize_return = namedtuple('optimize', 'converged X U')
        self.model_return = namedtuple('descent_model', 'Q R S')
        self.descent_return = namedtuple('calc_descent_direction', 'Kproj dX dU Q R S')
        self.armijo_search_return = namedtuple('armijo_search', 'nX nU cost1')
        self.armijo_simulate_return = namedtuple('armijo_simulate', 'success nX nU')
        self.check_dcost_return = namedtuple('check_dcost', 'result error cost1 cost0 approx_dcost exact_dcost')
        self.check_ddcost_return = namedtuple('check_ddcost', 'result error cost1 cost0 approx_ddcost exact_ddcost')
        self._device = device
        self._lin = None      # batch over the horizon (k-parallel linearisation, deriv2 contraction)
        self._arm = None      # batch over the Armijo candidates

    # -- engines ---------------------------------------------------------------------------------
    def _dt(self):
        """Step sizes of the DSystem's time base, one per step (the reference takes any time vector, dsystem.py:229-274)."""
        return np.diff(np.asarray(self.dsys.time, dtype=float))

    def _lin_engine(self, n):
        if self._lin is None or self._lin.batch != n:
            if self._lin is not None:
                self._lin.close()
            self._lin = BatchMidpointVI(self.dsys.system, n, device=self._device)
        self._lin.refresh()
        return self._lin

    def _arm_engine(self, n):
        if self._arm is None or self._arm.batch != n:
            if self._arm is not None:
                self._arm.close()
            self._arm = BatchMidpointVI(self.dsys.system, n, device=self._device)
        self._arm.refresh()
        return self._arm

    # -- cost -------------------------------------------------------------------------------------
    def _vectorised_cost(self):
        """The array-at-once DCost.total / DCost.gradients are an extension of this package; any other cost object
        (a subclass overriding l / m / l_dx ..., or a duck-typed one) is evaluated step by step through the reference's
        DCost contract (dcost.py:30-118), like the reference's own loops (doptimizer.py:249-270)."""
        from .dcost import DCost
        return type(self.cost) is DCost

    def _cost_gradients(self, X, U):
        c = self.cost
        if self._vectorised_cost():
            return c.gradients(X, U)
        n = len(X)
        q = np.array([c.l_dx(X[k], U[k], k) for k in range(n - 1)] + [c.m_dx(X[-1])])
        r = np.array([c.l_du(X[k], U[k], k) for k in range(n - 1)])
        return q, r

    def _cost_total(self, X, U):
        """Cost of one trajectory, or of a stack of trajectories [M][N+1][nX] / [M][N][nU] (array of M costs)."""
        c = self.cost
        X, U = np.asarray(X), np.asarray(U)
        if self._vectorised_cost():
            return c.total(X, U)
        if X.ndim == 3:
            return np.array([self._cost_total(X[m], U[m]) for m in range(len(X))])
        return sum(c.l(X[k], U[k], k) for k in range(len(X) - 1)) + c.m(X[-1])

    def calc_cost(self, X, U):
        return float(self._cost_total(X, U))

    def calc_dcost(self, X, U, dX, dU):
        q, r = self._cost_gradients(X, U)
        return float(np.sum(q * dX) + np.sum(r * dU))

    def calc_ddcost(self, X, U, dX, dU, Q, R, S):
        dd = 0.0
        for k in range(len(X) - 1):
            dd += dX[k].dot(Q(k)).dot(dX[k]) + 2 * dX[k].dot(S(k)).dot(dU[k]) + dU[k].dot(R(k)).dot(dU[k])
        return dd + dX[-1].dot(Q(-1)).dot(dX[-1])

    # -- linearisation (k-parallel) -------------------------------------------------------------------
    def linearize(self, X, U):
        """A [N][nX][nX], B [N][nX][nU] about (X, U); leaves the N solved steps resident on the device."""
        ds = self.dsys
        N = len(X) - 1
        dts = self._dt()[:N]
        eng = self._lin_engine(N)
        eng.set_step_sizes(dts, by_trajectory=True)      # trajectory k of the horizon batch steps by t[k+1] - t[k]
        dt = dts[:, None, None]                            # ... and its A_k / B_k blocks carry 1/dt_k
        Q, p, _ = ds.split_state(X)
        u, rho = ds.split_input(U)
        eng.initialize_from_state(ds.time[0], Q[:-1], p[:-1])
        iters, status = eng.step(ds.time[0] + float(dts[0]), u if ds._nu else None, rho if ds._nrho else None,
                                 q2_hint=Q[1:, :eng.nd])
        if (status != 0).any():
            raise ConvergenceError("linearisation: DEL solve failed at k=%s" % np.nonzero(status)[0][:5])
        eng.calc_deriv1()
        d = dict((n, np.swapaxes(eng.deriv1(n), 1, 2)) for n in eng.D1_NAMES[:8])
        nX, nU = ds.nX, ds.nU
        A = np.zeros((N, nX, nX))
        B = np.zeros((N, nX, nU))
        A[:, ds._slice_Qd, ds._slice_Q] = d["q2_dq1"]
        A[:, ds._slice_Qd, ds._slice_p] = d["q2_dp1"]
        A[:, ds._slice_p, ds._slice_Q] = d["p2_dq1"]
        A[:, ds._slice_p, ds._slice_p] = d["p2_dp1"]
        A[:, ds._slice_v, ds._slice_Qk] = -np.eye(ds._nv)[None] / dt
        B[:, ds._slice_Qd, ds._slice_u] = d["q2_du1"]
        B[:, ds._slice_Qd, ds._slice_rho] = d["q2_dk2"]
        B[:, ds._slice_Qk, ds._slice_rho] = np.eye(ds._nrho)
        B[:, ds._slice_p, ds._slice_u] = d["p2_du1"]
        B[:, ds._slice_p, ds._slice_rho] = d["p2_dk2"]
        B[:, ds._slice_v, ds._slice_rho] = np.eye(ds._nrho)[None] / dt
        return A, B

    def calc_feedback_controller(self, X, U):
        A, B = self.linearize(X, U)
        Kproj = dlqr.solve_tv_lqr(A, B, self.Qproj, self.Rproj)[0]
        return Kproj, A, B

    # -- quadratic models ---------------------------------------------------------------------------
    def calc_steepest_model(self):
        Q, R = np.eye(self.dsys.nX), np.eye(self.dsys.nU)
        S = np.zeros((self.dsys.nX, self.dsys.nU))
        return self.model_return(lambda k: Q, lambda k: R, lambda k: S)

    def calc_quasi_model(self, X, U):
        c = self.cost
        n = len(X)
        Q = [c.l_dxdx(X[k], U[k], k) for k in range(n - 1)] + [c.m_dxdx(X[-1])]
        S = [c.l_dxdu(X[k], U[k], k) for k in range(n - 1)]
        R = [c.l_dudu(X[k], U[k], k) for k in range(n - 1)]
        return self.model_return(lambda k: Q[k], lambda k: R[k], lambda k: S[k])

    def calc_newton_model(self, X, U, A, B, K):
        """Second-order model incl. the dynamics' curvature (doptimizer.py:319-345).  Must follow
        linearize(X, U): the contractions reuse the solved steps resident in the horizon batch."""
        c, ds = self.cost, self.dsys
        n = len(X)
        q, r = self._cost_gradients(X, U)
        Z = np.zeros((n - 1, ds.nX))
        z = q[-1]
        for k in range(n - 2, -1, -1):
            Z[k] = z                       # the adjoint that multiplies f's second derivatives at step k
            z = q[k] - r[k].dot(K[k]) + z.dot(A[k] - B[k].dot(K[k]))
        hz = self._lin.deriv2_contract(Z)
        Q = [None] * n
        S = [None] * (n - 1)
        R = [None] * (n - 1)
        Q[-1] = c.m_dxdx(X[-1])
        for k in range(n - 1):
            xx, xu, uu = ds._split_hz(hz[k])
            Q[k] = c.l_dxdx(X[k], U[k], k) + xx
            S[k] = c.l_dxdu(X[k], U[k], k) + xu
            R[k] = c.l_dudu(X[k], U[k], k) + uu
        return self.model_return(lambda k: Q[k], lambda k: R[k], lambda k: S[k])

    def calc_descent_direction(self, X, U, method='steepest'):
        (Kproj, A, B) = self.calc_feedback_controller(X, U)
        q, r = self._cost_gradients(X, U)
        if method == 'steepest':
            (Q, R, S) = self.calc_steepest_model()
        elif method == 'quasi':
            (Q, R, S) = self.calc_quasi_model(X, U)
        elif method == 'newton':
            (Q, R, S) = self.calc_newton_model(X, U, A, B, Kproj)
        else:
            raise Exception("Invalid descent direction method: %r" % method)
        (K, C, P, b) = dlqr.solve_tv_lq(A, B, q, r, Q, S, R)
        dx0 = -np.linalg.solve(P, b) if self.optimize_ic else np.zeros((self.dsys.nX,))
        dX = np.zeros(X.shape)
        dU = np.zeros(U.shape)
        dX[0] = dx0
        for k in range(len(X) - 1):
            dU[k] = -K[k].dot(dX[k]) - C[k]
            dX[k + 1] = A[k].dot(dX[k]) + B[k].dot(dU[k])
        return self.descent_return(Kproj, dX, dU, Q, R, S)

    # -- line search (m-parallel) ----------------------------------------------------------------------
    def project_candidates(self, X, U, Kproj, dX, dU, lambdas):
        """Closed-loop rollouts of X + lam dX, U + lam dU for every lam at once.
        Returns (nX [M][N+1][nX], nU [M][N][nU], ok [M])."""
        ds = self.dsys
        M = len(lambdas)
        N = len(X) - 1
        lam = np.asarray(lambdas, dtype=float)[:, None, None]
        bX = X[None] + lam * dX[None]
        bU = U[None] + lam * dU[None]
        eng = self._arm_engine(M)
        Q0, p0, _ = ds.split_state(bX[:, 0, :])
        eng.initialize_from_state(ds.time[0], Q0, p0)
        nX, nU = eng.rollout_closed_loop(N, self._dt(), np.asarray(Kproj)[None], bX, bU, group_size=M)
        nX[:, 0, :] = bX[:, 0, :]          # X[0] = bX[0] by definition of the projection (dsystem.py:441)
        _, status = eng.status()
        return nX, nU, status == 0

    def armijo_search(self, X, U, Kproj, dX, dU):
        cost0 = self.calc_cost(X, U)
        dcost0 = self.calc_dcost(X, U, dX, dU)
        lambdas = self.armijo_beta ** np.arange(self.armijo_max_iterations)
        nX, nU, ok = self.project_candidates(X, U, Kproj, dX, dU, lambdas)
        costs = self._cost_total(nX, nU)
        for m in range(self.armijo_max_iterations):
            max_cost = cost0 + self.armijo_alpha * lambdas[m] * dcost0
            if not ok[m]:
                self.monitor.armijo_simulation_failure(m, nX[m], nU[m], None, None)
                continue
            self.monitor.armijo_evaluation(m, nX[m], nU[m], None, None, float(costs[m]), max_cost)
            if costs[m] < max_cost:
                return self.armijo_search_return(nX[m], nU[m], float(costs[m]))
        self.monitor.armijo_search_failure(X, U, dX, dU, cost0, dcost0, Kproj)
        raise ConvergenceError("Armijo Failed to Converge")

    # -- iteration --------------------------------------------------------------------------------------
    def step(self, iteration, X, U, method='steepest'):
        """One iteration of the projection-operator descent.  Semantics of the reference's ``DOptimizer.step``
        (doptimizer.py:462-506), written as a loop over the method ladder instead of its recursion:

        1. build the descent direction of ``method`` about (X, U) and its directional derivative ``dcost0``;
        2. a direction that is not a descent direction (dcost0 > 0) demotes the method (newton -> quasi -> steepest,
           ``select_fallback_method``) and step 1 is repeated;
        3. |dcost0| below ``descent_tolerance`` ends the optimisation: returns (True, X, U, dcost0, cost0) unchanged;
        4. otherwise the Armijo search picks the step length along the direction (every candidate projected by a
           closed-loop rollout, all candidates in one batch) and the new trajectory is returned with done = False.
        The monitor sees the same sequence of events as the reference's."""
        self.monitor.step_begin(iteration)
        while True:
            direction = self.calc_descent_direction(X, U, method)
            cost0 = self.calc_cost(X, U)
            dcost0 = self.calc_dcost(X, U, direction.dX, direction.dU)
            self.monitor.step_info(method, cost0, dcost0, X, U, direction.dX, direction.dU, direction.Kproj)
            if dcost0 <= 0:
                break
            demoted = self.select_fallback_method(iteration, method)     # raises for 'steepest': nothing left to try
            self.monitor.step_method_failure(method, cost0, dcost0, demoted)
            method = demoted
            self.monitor.step_begin(iteration)      # the reference re-enters step() here
        if abs(dcost0) < self.descent_tolerance:
            self.monitor.step_termination(cost0, dcost0)
            return self.step_return(True, X, U, dcost0, cost0)
        found = self.armijo_search(X, U, direction.Kproj, direction.dX, direction.dU)
        self.monitor.step_completed(method, found.cost1, found.nX, found.nU)
        return self.step_return(False, found.nX, found.nU, dcost0, found.cost1)

    def armijo_simulate(self, bX, bU, Kproj):
        """Project (bX, bU) like DSystem.project; reports failure instead of raising (doptimizer.py:405-428).  The
        projection is one closed-loop device rollout, so a failed one yields no partial trajectory."""
        try:
            nX, nU = self.dsys.project(np.asarray(bX, dtype=float), np.asarray(bU, dtype=float), Kproj)
        except ConvergenceError:
            return self.armijo_simulate_return(False, np.zeros((0,) + np.shape(bX)[1:]), np.zeros((0,) + np.shape(bU)[1:]))
        return self.armijo_simulate_return(True, nX, nU)

    # -- finite-difference validators of the descent model (doptimizer.py:621-674): both probes in one batch ------
    def _probe(self, X, U, Kproj, dX, dU, delta):
        nX, nU, ok = self.project_candidates(X, U, Kproj, dX, dU, [-delta, delta])
        if not ok.all():
            raise ConvergenceError("finite-difference check: projection failed")
        return nX, nU

    def check_dcost(self, X, U, method='steepest', delta=1e-6, tolerance=1e-5):
        (Kproj, dX, dU, Q, R, S) = self.calc_descent_direction(X, U, method)
        exact = self.calc_dcost(X, U, dX, dU)
        nX, nU = self._probe(X, U, Kproj, dX, dU, delta)
        cost0, cost1 = self.calc_cost(nX[0], nU[0]), self.calc_cost(nX[1], nU[1])
        approx = (cost1 - cost0) / (2 * delta)
        error = approx - exact
        return self.check_dcost_return(abs(error) <= tolerance, error, cost1, cost0, approx, exact)

    def check_ddcost(self, X, U, method='steepest', delta=1e-6, tolerance=1e-5):
        (Kproj, dX, dU, Q, R, S) = self.calc_descent_direction(X, U, method)
        if method != 'newton':
            (Q, R, S) = self.calc_descent_direction(X, U, 'newton')[-3:]
        exact = self.calc_ddcost(X, U, dX, dU, Q, R, S)
        nX, nU = self._probe(X, U, Kproj, dX, dU, delta)
        dcosts = []
        for i in range(2):
            (A, B) = self.linearize(nX[i], nU[i])
            (ndX, ndU) = self.dsys.dproject(A, B, dX, dU, Kproj)
            dcosts.append(self.calc_dcost(nX[i], nU[i], ndX, ndU))
        approx = (dcosts[1] - dcosts[0]) / (2 * delta)
        error = approx - exact
        return self.check_ddcost_return(abs(error) <= tolerance, error, dcosts[1], dcosts[0], approx, exact)

    # -- cost along a descent direction (doptimizer.py:569-617) ------------------------------------------------------------
    def descent_curves(self, X, U, method='steepest', points=40):
        """The data of the reference's descent plot: the projected cost g(z) = cost(P(X + z dX, U + z dU)) on `points` values of
        z in [-0.1, 1.01] plus the Armijo steps beta^m, m < 20, the quadratic model cost + z dcost + z^2 ddcost / 2 and the
        Armijo bound cost + alpha z dcost.  All projections are ONE batch of closed-loop rollouts on the device.  Returns a dict
        of arrays (differences to the cost at z = 0; NaN where a projection did not converge)."""
        (Kproj, dX, dU, Q, R, S) = self.calc_descent_direction(X, U, method)
        armijo_z = np.sort(self.armijo_beta ** np.arange(20.0))
        z = np.sort(np.concatenate((np.linspace(-0.1, 1.01, points), armijo_z)))
        cost = self.calc_cost(X, U)
        dcost = self.calc_dcost(X, U, dX, dU)
        ddcost = self.calc_ddcost(X, U, dX, dU, Q, R, S)
        nX, nU, ok = self.project_candidates(X, U, Kproj, dX, dU, z)
        true = np.array([self.calc_cost(nX[i], nU[i]) if ok[i] else np.nan for i in range(len(z))])
        pick = np.searchsorted(z, armijo_z)
        return {"z": z, "true": true - cost, "model": dcost * z + 0.5 * ddcost * z * z, "armijo_z": armijo_z,
                "armijo": true[pick] - cost, "required": self.armijo_alpha * z * dcost, "cost": cost, "dcost": dcost, "ddcost": ddcost}

    def descent_plot(self, X, U, method='steepest', points=40, legend=True):
        """Plot descent_curves() with matplotlib (same four curves and labels as the reference's plot)."""
        try:
            from matplotlib import pyplot
        except ImportError:
            raise RuntimeError("Importing matplotlib failed. Cannot create plot.")
        d = self.descent_curves(X, U, method, points)
        pyplot.plot(d["z"], d["model"], '-,', linewidth=2.0, color='blue', label='Modeled Cost')
        pyplot.plot(d["z"], d["true"], '.-', linewidth=1.0, color='black', label='True Cost')
        pyplot.plot(d["armijo_z"], d["armijo"], 'o', color='gray', label='Armijo Evaluations')
        pyplot.plot(d["z"], d["required"], '-.', color='black', label='Required Cost Improvement')
        if legend:
            pyplot.legend(loc=0)
        pyplot.title('Cost along descent direction for method: "%s".' % method)
        pyplot.xlabel('z')
        pyplot.ylabel(r'$\Delta$ cost')

    def select_method(self, iteration):
        return self.first_method if iteration < self.first_method_iterations else self.second_method

    def select_fallback_method(self, iteration, current_method):
        if current_method == 'newton':
            return 'quasi'
        if current_method == 'quasi':
            return 'steepest'
        raise Exception("Derivative of cost is positive for steepest descent.")

    def optimize(self, X, U, max_steps=50):
        X = np.array(X)
        U = np.array(U)
        self.monitor.optimize_begin(X, U)
        converged, cost = False, None
        for i in range(max_steps):
            (converged, X, U, dcost, cost) = self.step(i, X, U, self.select_method(i))
            if converged:
                break
        self.monitor.optimize_end(converged, X, U, cost)
        return self.optimize_return(converged, X, U)
