"""``System``: the container that owns the frame tree, coordinates, potentials,
forces and constraints, and derives the integer topology tables from them.

Mirrors the model-building surface of the reference's ``trep.System``
(/root/reference/trep/system.py:22-158, 283-291, 306-660) and its "structure
sync" (system.py:672-771, frame.py:658-691).  The tables computed by ``_sync``
(frame order, configs = dyn + kin, config_gen, index/k_index, masses,
cache_index, config.masses) are exactly what the device library consumes
(``trep_amd.descriptor``) and are checked bit-for-bit against tables dumped from
the reference (tests/test_topology.py).

Next to the hot path the class also offers the continuous dynamics -- ``f``, ``lambda_``, their first
derivatives (analytic kernels) and second derivatives (the first-derivative kernel on dual numbers: exact), energies
and Lagrangian derivatives -- ``satisfy_constraints`` (SLSQP on the host, gradients from the device) and the
trajectory ``.mat`` files (``save_trajectory`` / ``load_trajectory``).
"""
import numpy as np

from .config import Config, Input
from .frame import Frame, WORLD


def dynamics_deriv2_forward(deriv1_forward, q, dq, u, ddqk):
    """Second derivatives of the continuous dynamics from a batched FORWARD-MODE evaluator of the analytic first derivatives.

    deriv1_forward(Q [M][nq], dQ [M][nq], U [M][nu], ddK [M][nk], seed [M]) -> {"f_dq": [M][nd][nq], "f_ddq", "f_dddk" [M][nd][nk],
    "f_du", "lambda_dq" [M][nc][nq], ...}: for state m the exact derivative of every first-derivative array along input variable
    seed[m] (numbered q | dq | ddq_k | u) -- BatchMidpointVI.dynamics_deriv1(..., seeds=(seed,)), i.e. the first-derivative kernel run
    on dual numbers (csrc/dual.hpp); the tests also drive it with the host emulation of that kernel.  One state per variable of
    (q, dq, u): M = 2 nq + nu states in one call, no step size.  Returns the reference's fourteen arrays, [first variable][second
    variable][output] (system.py:982-1078)."""
    q, dq, u, ddqk = (np.asarray(a, dtype=float) for a in (q, dq, u, ddqk))
    nq, nu, nk = len(q), len(u), len(ddqk)
    nv = 2 * nq + nu
    seed = np.concatenate([np.arange(2 * nq), 2 * nq + nk + np.arange(nu)]).astype(np.int32)
    rep = lambda a: np.repeat(a[None], nv, axis=0)
    d = deriv1_forward(rep(q), rep(dq), rep(u), rep(ddqk), seed)

    def part(name, lo, hi):
        """d(name[output][var1]) / d x_v for v in [lo, hi): -> [var1][v][output]"""
        return np.ascontiguousarray(np.transpose(np.asarray(d[name])[lo:hi], (2, 0, 1)))
    out = {}
    for pre in ("f", "lambda"):
        out[pre + "_dqdq"] = part(pre + "_dq", 0, nq)
        out[pre + "_ddqdq"] = part(pre + "_ddq", 0, nq)
        out[pre + "_ddqddq"] = part(pre + "_ddq", nq, 2 * nq)
        out[pre + "_dddkdq"] = part(pre + "_dddk", 0, nq)
        out[pre + "_dudq"] = part(pre + "_du", 0, nq)
        out[pre + "_duddq"] = part(pre + "_du", nq, 2 * nq)
        out[pre + "_dudu"] = part(pre + "_du", 2 * nq, 2 * nq + nu)
    return out


class System(object):
    def __init__(self):
        self._dyn_configs = tuple()
        self._kin_configs = tuple()
        self._potentials = tuple()
        self._forces = tuple()
        self._inputs = tuple()
        self._constraints = tuple()
        self._frames = tuple()
        self._configs = tuple()
        self._masses = tuple()
        self._time = 0.0
        self._hold = 0
        self._dirty = True
        self._structure_version = 0
        self._structure_changed_funcs = []
        self._world_frame = Frame(self, WORLD, None, name="World")

    def __repr__(self):
        return "<System %d configs, %d frames, %d potentials, %d constraints, %d forces, %d inputs>" % (
            len(self.configs), len(self.frames), len(self.potentials),
            len(self.constraints), len(self.forces), len(self.inputs))

    # -- registration (called by the model objects' constructors) -------------
    def _register_config(self, config):
        if config.kinematic:
            self._kin_configs += (config,)
        else:
            self._dyn_configs += (config,)
        self._structure_changed()

    def _register_input(self, finput):
        self._inputs += (finput,)
        self._structure_changed()

    def _add_potential(self, potential):
        self._potentials += (potential,)
        self._structure_changed()

    def _add_force(self, force):
        self._forces += (force,)
        self._structure_changed()

    def _add_constraint(self, constraint):
        self._constraints += (constraint,)
        self._structure_changed()

    # -- structure sync -------------------------------------------------------
    def hold_structure_changes(self):
        self._hold += 1

    def resume_structure_changes(self):
        if self._hold == 0:
            raise Exception("System.resume_structure_changes() called when _hold_structure_changes is 0")
        self._hold -= 1
        if self._hold == 0:
            self._structure_changed()

    def add_structure_changed_func(self, function):
        self._structure_changed_funcs.append(function)

    def _structure_changed(self):
        self._dirty = True
        self._structure_version += 1
        if self._hold == 0:
            for func in list(self._structure_changed_funcs):
                func()

    def _sync(self):
        """Recompute every derived table (lazily, once per structural change)."""
        if not self._dirty:
            return
        self._dirty = False
        self._frames = tuple(self._world_frame.flatten_tree())
        self._configs = self._dyn_configs + self._kin_configs
        nq = len(self._configs)

        # config_gen = number of variable frames strictly above the driven frame
        # on its path from the world frame; configs that drive no frame keep nq.
        for config in self._configs:
            config._config_gen = nq
        stack = [(self._world_frame, 0)]
        while stack:
            frame, depth = stack.pop()
            if frame._config is not None:
                frame._config._config_gen = depth
                depth += 1
            for child in frame._children:
                stack.append((child, depth))

        for i, config in enumerate(self._configs):
            config._index = i
            config._k_index = -1
        for i, config in enumerate(self._kin_configs):
            config._k_index = i
        for i, constraint in enumerate(self._constraints):
            constraint._index = i
        for i, finput in enumerate(self._inputs):
            finput._index = i

        self._masses = tuple(f for f in self._frames
                             if f._mass != 0.0 or f._Ixx != 0.0 or f._Iyy != 0.0 or f._Izz != 0.0)

        # cache_index: configs on the path world -> frame, padded with None to nq+1.
        for frame in self._frames:
            path = []
            f = frame
            while f is not None:
                if f._config is not None:
                    path.append(f._config)
                f = f._parent
            path.reverse()
            frame._cache_size = len(path)
            frame._cache_index = tuple(path + [None] * (nq + 1 - len(path)))

        for config in self._configs:
            config._masses = tuple(f for f in self._masses if config in f._cache_index[:f._cache_size])

    # -- sizes ------------------------------------------------------------------
    nQ = property(lambda self: len(self.configs))
    nQd = property(lambda self: len(self._dyn_configs))
    nQk = property(lambda self: len(self._kin_configs))
    nu = property(lambda self: len(self._inputs))
    nc = property(lambda self: len(self._constraints))

    @property
    def t(self):
        return self._time

    @t.setter
    def t(self, value):
        self._time = value

    # -- collections --------------------------------------------------------------
    world_frame = property(lambda self: self._world_frame)
    dyn_configs = property(lambda self: self._dyn_configs)
    kin_configs = property(lambda self: self._kin_configs)
    potentials = property(lambda self: self._potentials)
    forces = property(lambda self: self._forces)
    inputs = property(lambda self: self._inputs)
    constraints = property(lambda self: self._constraints)

    @property
    def frames(self):
        self._sync()
        return self._frames

    @property
    def configs(self):
        self._sync()
        return self._configs

    @property
    def masses(self):
        self._sync()
        return self._masses

    # -- lookup -------------------------------------------------------------------
    @staticmethod
    def _get_object(identifier, objtype, array):
        if identifier is None:
            return None
        if isinstance(identifier, objtype):
            return identifier
        if isinstance(identifier, (int, np.integer)):
            return array[identifier]
        if isinstance(identifier, str):
            for item in array:
                if item.name == identifier:
                    return item
            raise KeyError("%s with name '%s' not found" % (objtype, identifier))
        raise TypeError()

    def get_frame(self, identifier):
        return self._get_object(identifier, Frame, self.frames)

    def get_config(self, identifier):
        return self._get_object(identifier, Config, self.configs)

    def get_input(self, identifier):
        return self._get_object(identifier, Input, self.inputs)

    def get_potential(self, identifier):
        from .dynamics import Potential
        return self._get_object(identifier, Potential, self.potentials)

    def get_force(self, identifier):
        from .dynamics import Force
        return self._get_object(identifier, Force, self.forces)

    def get_constraint(self, identifier):
        from .dynamics import Constraint
        return self._get_object(identifier, Constraint, self.constraints)

    def import_frames(self, children):
        self._world_frame.import_frames(children)

    def export_frames(self, system_name='system', frames_name='frames', tab_size=4):
        """Python source that rebuilds this system's frame tree (system.py:292-305)."""
        txt = '#' * 80 + '\n# Frame tree definition generated by System.export_frames()\n\n'
        txt += 'from trep import const_se3, rx, ry, rz, tx, ty, tz\n'
        txt += '%s = [\n' % frames_name
        txt += ',\n'.join(child.export_frames(1, tab_size) for child in self._world_frame.children) + '\n'
        txt += ' ' * tab_size + ']\n'
        txt += '%s.import_frames(%s)\n' % (system_name, frames_name)
        return txt + '#' * 80 + '\n'


    # -- state vectors ---------------------------------------------------------------
    @staticmethod
    def _assign(items, attr, value, lookup):
        if isinstance(value, (int, float)):
            for it in items:
                setattr(it, attr, value)
        elif isinstance(value, dict):
            for name, v in value.items():
                setattr(lookup(name), attr, v)
        else:
            for it, v in zip(items, value):
                setattr(it, attr, v)

    def _state_prop(collection, attr, lookup="get_config"):
        def getter(self):
            return np.array([getattr(x, attr) for x in getattr(self, collection)], dtype=float)

        def setter(self, value):
            self._assign(getattr(self, collection), attr, value, getattr(self, lookup))
        return property(getter, setter)

    q = _state_prop("configs", "q")
    dq = _state_prop("configs", "dq")
    ddq = _state_prop("configs", "ddq")
    qd = _state_prop("dyn_configs", "q")
    dqd = _state_prop("dyn_configs", "dq")
    ddqd = _state_prop("dyn_configs", "ddq")
    qk = _state_prop("kin_configs", "q")
    dqk = _state_prop("kin_configs", "dq")
    ddqk = _state_prop("kin_configs", "ddq")
    u = _state_prop("inputs", "u", "get_input")
    del _state_prop

    def satisfy_constraints(self, tolerance=1e-10, verbose=False, keep_kinematic=False, constant_q_list=None):
        """Move the configuration to the nearest one (least squares) that satisfies the holonomic constraints
        (host-side setup, same approach and arguments as trep/system.py:158-214: SLSQP on |q - q0|^2 subject to
        h(q) = 0 with analytic constraint gradients).  Velocities are set to zero.  Returns the new configuration."""
        import scipy.optimize
        self.dq = 0
        if constant_q_list:
            fixed = set(self.get_config(c).name for c in constant_q_list)
            free = [c for c in self.configs if c.name not in fixed]
        elif keep_kinematic:
            free = list(self.dyn_configs)
        else:
            free = list(self.configs)
        q0 = np.array([c.q for c in free], dtype=float)

        def put(q):
            for c, v in zip(free, q):
                c.q = v

        def f_eqcons(q):
            put(q)
            return np.array([c.h() for c in self.constraints])

        def fprime_eqcons(q):
            put(q)
            return np.array([[c.h_dq(cfg) for cfg in free] for c in self.constraints])

        (q_opt, fx, its, imode, smode) = scipy.optimize.fmin_slsqp(
            lambda q: float((q - q0).dot(q - q0)), q0, f_eqcons=f_eqcons, fprime=lambda q: 2.0 * (q - q0),
            fprime_eqcons=fprime_eqcons, acc=tolerance, iter=100 * self.nQ, iprint=1 if verbose else 0, full_output=True)
        if imode != 0:
            raise Exception("Minimization failed: %s" % smode)
        put(q_opt)
        return self.q

    def minimize_potential_energy(self, tolerance=1e-10, verbose=False, keep_kinematic=False, constant_q_list=None):
        """Move to a nearby configuration that minimises the potential energy subject to the holonomic constraints
        (equilibrium search; same approach and arguments as trep/system.py:215-272: SLSQP on V(q) with h(q) = 0).
        V comes from the energy kernel; its gradient from central differences evaluated as ONE batch of 2n states
        (the reference uses -L_dq).  Velocities are set to zero.  Returns the new configuration."""
        import scipy.optimize
        from .midpointvi import BatchMidpointVI
        self.dq = 0
        if constant_q_list:
            fixed = set(self.get_config(c).name for c in constant_q_list)
            free = [c for c in self.configs if c.name not in fixed]
        elif keep_kinematic:
            free = list(self.dyn_configs)
        else:
            free = list(self.configs)
        q0 = np.array([c.q for c in free], dtype=float)
        idx = np.array([c.index for c in free], dtype=int)
        n = len(free)
        grad_engine = BatchMidpointVI(self, 2 * n)
        delta = 1e-6

        def put(q):
            for c, v in zip(free, q):
                c.q = v

        def func(q):
            put(q)
            return float(self._energies()[1])

        def fprime(q):
            put(q)
            Q = np.tile(self.q, (2 * n, 1))
            Q[np.arange(n), idx] += delta
            Q[n + np.arange(n), idx] -= delta
            V = grad_engine.energy(Q, np.zeros_like(Q))[:, 1]
            return (V[:n] - V[n:]) / (2 * delta)

        def f_eqcons(q):
            put(q)
            return np.array([c.h() for c in self.constraints])

        def fprime_eqcons(q):
            put(q)
            return np.array([[c.h_dq(cfg) for cfg in free] for c in self.constraints]).reshape(len(self.constraints), n)

        try:
            (q_opt, fx, its, imode, smode) = scipy.optimize.fmin_slsqp(
                func, q0, f_eqcons=f_eqcons if self.constraints else None, fprime=fprime,
                fprime_eqcons=fprime_eqcons if self.constraints else None, acc=tolerance, iter=100 * self.nQ,
                iprint=1 if verbose else 0, full_output=True)
        finally:
            grad_engine.close()
        if imode != 0:
            raise Exception("Minimization failed: %s" % smode)
        put(q_opt)
        return self.q

    # -- continuous dynamics (system.py:951-959, 1018-1024 of the reference; calc_dynamics system.c:749-893) --------
    def _dynamics_engine(self):
        from .midpointvi import BatchMidpointVI
        eng = getattr(self, "_dyn_engine", None)
        if eng is None or self._dyn_engine_version != self._structure_version:
            if eng is not None:
                eng.close()
            eng = self._dyn_engine = BatchMidpointVI(self, 1)
            self._dyn_engine_version = self._structure_version
        return eng

    def _dynamics(self):
        """Accelerations and constraint forces at the current (q, dq, u, ddqk): one launch of the dynamics kernel
        on a batch of one.  Like the reference, the result is also stored in the dynamic configs' ddq."""
        eng = self._dynamics_engine()
        ddq, lam, status = eng.dynamics(self.q[None], self.dq[None], self.u[None], self.ddqk[None])
        if status[0] != 0:
            raise ValueError("singular inertia or constraint matrix")   # LU_decomp failure in the reference
        self.ddqd = ddq[0]
        return ddq[0], lam[0]

    def f(self, q=None):
        """ddq of the dynamic configs (all of them, or of config ``q``)."""
        ddq = self._dynamics()[0]
        if q is None:
            return ddq
        assert not q.kinematic
        return float(ddq[q.index])

    def lambda_(self, constraint=None):
        lam = self._dynamics()[1]
        return lam if constraint is None else float(lam[constraint.index])

    def _energies(self):
        self._dynamics_engine()
        return self._dyn_engine.energy(self.q[None], self.dq[None])[0]

    def total_energy(self):
        """Kinetic plus potential energy at the current state (system.py:844-846)."""
        T, V = self._energies()
        return float(T + V)

    def L(self):
        """The Lagrangian at the current state (system.py:848-850)."""
        T, V = self._energies()
        return float(T - V)

    def _lagrangian(self):
        d = self._dynamics_engine().lagrangian(self.q[None], self.dq[None])
        return dict((k, v[0]) for k, v in d.items())

    def L_dq(self, q1):
        """dL/dq1 at the current state (system.py:852-858)."""
        return float(self._lagrangian()["L_dq"][q1.index])

    def L_ddq(self, dq1):
        return float(self._lagrangian()["L_ddq"][dq1.index])

    def L_dqdq(self, q1, q2):
        return float(self._lagrangian()["L_dqdq"][q1.index, q2.index])

    def L_ddqdq(self, dq1, q2):
        return float(self._lagrangian()["L_ddqdq"][dq1.index, q2.index])

    def L_ddqddq(self, dq1, dq2):
        return float(self._lagrangian()["L_ddqddq"][dq1.index, dq2.index])

    # -- third- and fourth-order derivatives of the Lagrangian (system.py:869-949; System_L_dqdqdq ... L_ddqddqdqdq, system.c:204-557)
    def _L_higher(self, name, i, j, wrt):
        """d^n (name[i][j]) / dq_wrt[0] (dq_wrt[1]) with name in L_dqdq / L_ddqdq / L_ddqddq, n = len(wrt) in (1, 2): the
        reference reads these from its third- and fourth-order frame tables; here the ANALYTIC second-order arrays of the
        Lagrangian kernel are differentiated exactly -- the same kernel on dual numbers (csrc/dual.hpp: one direction for the
        third order, two nested ones for the fourth), no step size.  Agreement with the reference: rounding (1e-13)."""
        from .midpointvi import BatchMidpointVI
        eng = getattr(self, "_lag_engine", None)
        if eng is None or self._lag_engine_version != self._structure_version:
            if eng is not None:
                eng.close()
            eng = self._lag_engine = BatchMidpointVI(self, 1)
            self._lag_engine_version = self._structure_version
        seeds = tuple(np.array([k], dtype=np.int32) for k in wrt)      # configuration variables come first in the numbering
        return float(eng.lagrangian(self.q[None], self.dq[None], seeds=seeds)[name][0, i, j])

    def L_dqdqdq(self, q1, q2, q3):
        value = self._L_higher("L_dqdq", q1.index, q2.index, (q3.index,))
        if q1 is q2 and q2 is q3:
            # NonlinearConfigSpring: the reference's V_dqdqdq is -y'' (-m) m (nonlinear_config_spring.c:56-60), the derivative of
            # the kernel's V_dqdq is -y'' m m; L holds -V, so the reference's value is the differenced one minus twice its V_dqdqdq
            from .dynamics import NonlinearConfigSpring
            from . import element_queries as _eq
            for p_ in self.potentials:
                if isinstance(p_, NonlinearConfigSpring) and p_.config is q1:
                    value -= 2.0 * _eq.nonlinear_config_spring(p_, (q1, q1, q1))
        return value

    def L_ddqdqdq(self, dq1, q2, q3):
        return self._L_higher("L_ddqdq", dq1.index, q2.index, (q3.index,))

    def L_ddqdqdqdq(self, dq1, q2, q3, q4):
        return self._L_higher("L_ddqdq", dq1.index, q2.index, (q3.index, q4.index))

    def L_ddqddqdq(self, dq1, dq2, q3):
        return self._L_higher("L_ddqddq", dq1.index, dq2.index, (q3.index,))

    def L_ddqddqdqdq(self, dq1, dq2, q3, q4):
        return self._L_higher("L_ddqddq", dq1.index, dq2.index, (q3.index, q4.index))

    def _dynamics_deriv1(self):
        self._dynamics()                      # engine + the reference's side effect on Config.ddq
        d, status = self._dyn_engine.dynamics_deriv1(self.q[None], self.dq[None], self.u[None], self.ddqk[None])
        if status[0] != 0:
            raise ValueError("singular inertia or constraint matrix")
        return dict((k, v[0]) for k, v in d.items())

    def _dyn_d1_accessor(name, out_kind, var_kind):
        def pick(obj, kind):
            if obj is None:
                return slice(None)
            if kind == "d":
                assert not obj.kinematic
                return obj.index
            return obj.k_index if kind == "k" else obj.index
        def accessor(self, out=None, var=None):
            """[output][derivative variable] like the reference (system.py:961-980, 1026-1044); objects select entries."""
            return np.array(self._dynamics_deriv1()[name][pick(out, out_kind), pick(var, var_kind)])
        accessor.__name__ = name
        return accessor

    f_dq = _dyn_d1_accessor("f_dq", "d", "q")
    f_ddq = _dyn_d1_accessor("f_ddq", "d", "q")
    f_dddk = _dyn_d1_accessor("f_dddk", "d", "k")
    f_du = _dyn_d1_accessor("f_du", "d", "u")
    lambda_dq = _dyn_d1_accessor("lambda_dq", "c", "q")
    lambda_ddq = _dyn_d1_accessor("lambda_ddq", "c", "q")
    lambda_dddk = _dyn_d1_accessor("lambda_dddk", "c", "k")
    lambda_du = _dyn_d1_accessor("lambda_du", "c", "u")
    del _dyn_d1_accessor

    # -- second derivatives of the continuous dynamics (system.py:982-1078; calc_dynamics_deriv2, system.c:1301-2029) ----
    def _dynamics_deriv2(self):
        """All fourteen second-derivative arrays of ddq = f(q, dq, u, ddq_k) and lambda at the current state, laid out
        like the reference's ([first variable][second variable][output]).

        The reference assembles them from fourth-order Lagrangian tables (M_dqdq is nq^4 entries, system.c:1301-2029).  Here
        they are the exact derivatives of the analytic first-derivative arrays (MODE_DYN_DERIV1) along each of q, dq and u:
        the first-derivative kernel run on dual numbers (csrc/dual.hpp), one input variable per trajectory, 2 nq + nu
        trajectories in one launch.  No step size and no truncation: agreement with the reference is rounding (1e-13
        relative to each array's largest entry on the test systems, tested at 1e-10).
        Systems with a LinearDamper: the reference's own f_ddqdq is inconsistent with its first derivatives there
        (lineardamper.c:88 uses length_dq where length_dqdq is meant); _apply_reference_conventions carries that over."""
        from .midpointvi import BatchMidpointVI
        self._dynamics()                       # the reference's side effect on Config.ddq; builds the engine
        nv = 2 * self.nQ + self.nu
        eng = getattr(self, "_dyn2_engine", None)
        if eng is None or self._dyn2_engine_version != self._structure_version or eng.batch != nv:
            if eng is not None:
                eng.close()
            eng = self._dyn2_engine = BatchMidpointVI(self, nv)
            self._dyn2_engine_version = self._structure_version
        def deriv1_forward(Q, dQ, U, ddK, seed):
            d, status = eng.dynamics_deriv1(Q, dQ, U, ddK, seeds=(seed,))
            if (status != 0).any():
                raise ValueError("singular inertia or constraint matrix")
            return d
        out = dynamics_deriv2_forward(deriv1_forward, self.q, self.dq, self.u, self.ddqk)
        if getattr(self, "reference_conventions", True):      # False: the derivatives consistent with this library's own first derivatives
            self._apply_reference_conventions(out)
        return out

    def _apply_reference_conventions(self, out, mass_matrix=None):
        """The reference evaluates two element derivatives with its own conventions, and calc_dynamics_deriv2 carries them into
        f_* / lambda_* (system.c:1301-2029); the discrete path here (MODE_DERIV2Z) reproduces both, so the continuous one does too:

        * ``LinearDamper`` F_ddqdq(q; dq1, q2) = -c (v_ddq1dq2 x_q + v_ddq1 * length_dq(q2)) -- lineardamper.c:88 has length_dq(q2)
          where the derivative of the first-order term has length_dqdq(q, q2);
        * ``NonlinearConfigSpring`` V_dqdqdq = -y'' (-m) m (nonlinear_config_spring.c:56-60; the derivative of V_dqdq is -y'' m m).

        A force second derivative enters the dynamics' second derivatives only through D (system.c:775-795) and D only linearly:
        d(ddq) = M^-1 (dD + Ad^T d(lambda)), d(lambda) = -(Ad M^-1 Ad^T)^-1 Ad M^-1 dD (system.c:840-892).  So the reference's arrays
        are the consistent ones (derivatives of the analytic first derivatives) plus that linear response to the difference
        between the reference's element derivative and the consistent one."""
        from .dynamics import LinearDamper, NonlinearConfigSpring
        from . import element_queries as _eq
        dampers = [f for f in self.forces if isinstance(f, LinearDamper)]
        springs = [p for p in self.potentials if isinstance(p, NonlinearConfigSpring)]
        if not dampers and not springs:
            return
        nq, nd, nc = self.nQ, self.nQd, self.nc
        M = (self._lagrangian()["L_ddqddq"] if mass_matrix is None else np.asarray(mass_matrix))[:nd, :nd]   # the tests pass the emulated kernel's
        Minv = np.linalg.inv(M)
        if nc:
            Ad = np.array([[c.h_dq(q) for q in self.dyn_configs] for c in self.constraints])
            Gl = -np.linalg.solve(Ad.dot(Minv).dot(Ad.T), Ad.dot(Minv))        # d lambda / dD  [nc][nd]
            Gf = Minv.dot(np.eye(nd) + Ad.T.dot(Gl))                            # d ddq / dD     [nd][nd]
        else:
            Gl, Gf = np.zeros((0, nd)), Minv
        configs = self.configs
        for f in dampers:
            used = [q for q in configs if f._on(q)]
            x_q = dict((q.index, _eq.length_dq(f, q)) for q in used)
            for dq1 in used:
                for q2 in used:
                    dD = np.zeros(nd)
                    for q in used:
                        if q.index < nd:
                            dD[q.index] = -f.c * x_q[dq1.index] * (x_q[q2.index] - _eq.length_dqdq(f, q, q2))
                    out["f_ddqdq"][dq1.index, q2.index] += Gf.dot(dD)
                    if nc:
                        out["lambda_ddqdq"][dq1.index, q2.index] += Gl.dot(dD)
        for p_ in springs:
            k = p_.config.index
            if k >= nd:
                continue          # a potential on a kinematic config exerts no generalized force on the dynamic ones
            ref3 = _eq.nonlinear_config_spring(p_, (p_.config, p_.config, p_.config))      # the reference's V_dqdqdq
            dD = np.zeros(nd)
            dD[k] = -2.0 * ref3           # D holds -V_dq: reference minus consistent third derivative = 2 * reference's value
            out["f_dqdq"][k, k] += Gf.dot(dD)
            if nc:
                out["lambda_dqdq"][k, k] += Gl.dot(dD)

    def _dyn_d2_accessor(name, out_kind, kind1, kind2):
        def pick(obj, kind):
            if obj is None:
                return slice(None)
            if kind == "d":
                assert not obj.kinematic
                return obj.index
            return obj.k_index if kind == "k" else obj.index
        def accessor(self, out=None, var1=None, var2=None):
            """[first variable][second variable][output] like the reference (system.py:982-1016, 1046-1078)."""
            return np.array(self._dynamics_deriv2()[name][pick(var1, kind1), pick(var2, kind2), pick(out, out_kind)])
        accessor.__name__ = name
        return accessor

    f_dqdq = _dyn_d2_accessor("f_dqdq", "d", "q", "q")
    f_ddqdq = _dyn_d2_accessor("f_ddqdq", "d", "q", "q")
    f_ddqddq = _dyn_d2_accessor("f_ddqddq", "d", "q", "q")
    f_dddkdq = _dyn_d2_accessor("f_dddkdq", "d", "k", "q")
    f_dudq = _dyn_d2_accessor("f_dudq", "d", "u", "q")
    f_duddq = _dyn_d2_accessor("f_duddq", "d", "u", "q")
    f_dudu = _dyn_d2_accessor("f_dudu", "d", "u", "u")
    lambda_dqdq = _dyn_d2_accessor("lambda_dqdq", "c", "q", "q")
    lambda_ddqdq = _dyn_d2_accessor("lambda_ddqdq", "c", "q", "q")
    lambda_ddqddq = _dyn_d2_accessor("lambda_ddqddq", "c", "q", "q")
    lambda_dddkdq = _dyn_d2_accessor("lambda_dddkdq", "c", "k", "q")
    lambda_dudq = _dyn_d2_accessor("lambda_dudq", "c", "u", "q")
    lambda_duddq = _dyn_d2_accessor("lambda_duddq", "c", "u", "q")
    lambda_dudu = _dyn_d2_accessor("lambda_dudu", "c", "u", "u")
    del _dyn_d2_accessor

    # -- numeric validators (system.py:1080-1203): func() -> float or array, func_d(config) -> its derivative -------
    def _test_derivative(self, attr, func, func_d, delta, tolerance, verbose, test_name):
        x0 = getattr(self, attr)
        failed = total = 0
        for q in self.configs:
            setattr(self, attr, x0)
            exact = func_d(q)
            x = x0.copy(); x[q.index] -= delta
            setattr(self, attr, x)
            y0 = func()
            x = x0.copy(); x[q.index] += delta
            setattr(self, attr, x)
            y1 = func()
            approx = (y1 - y0) / (2 * delta)
            error = np.linalg.norm(exact - approx)
            total += 1
            if not (error <= tolerance):
                failed += 1
                if verbose:
                    print("Test '%s' failed for the %s derivative of %r: error %g > %g" % (test_name, attr, q, error, tolerance))
        if verbose:
            print("%d tests passing." % total if not failed else "%d/%d tests FAILED." % (failed, total))
        setattr(self, attr, x0)
        return not failed

    def test_derivative_dq(self, func, func_dq, delta=1e-6, tolerance=1e-7, verbose=False, test_name='<unnamed>'):
        """Central-difference check of a derivative w.r.t. the configuration values."""
        return self._test_derivative("q", func, func_dq, delta, tolerance, verbose, test_name)

    def test_derivative_ddq(self, func, func_ddq, delta=1e-6, tolerance=1e-7, verbose=False, test_name='<unnamed>'):
        """Central-difference check of a derivative w.r.t. the configuration velocities."""
        return self._test_derivative("dq", func, func_ddq, delta, tolerance, verbose, test_name)

    def set_state(self, q=None, dq=None, u=None, ddqk=None, t=None):
        if q is not None:
            self.q = q
        if dq is not None:
            self.dq = dq
        if u is not None:
            self.u = u
        if ddqk is not None:
            self.ddqk = ddqk
        if t is not None:
            self.t = t


def save_trajectory(filename, system, t, Q=None, p=None, v=None, u=None, rho=None):
    """Write a trajectory to a MATLAB file, same variables as the reference (system.py:1209-1230):
    ``time``, the given arrays of ``Q p v u rho``, and ``*_index`` cell arrays with the config / input names."""
    import scipy.io
    data = {"time": np.array(t)}
    for key, value in (("Q", Q), ("p", p), ("v", v), ("u", u), ("rho", rho)):
        if value is not None:
            data[key] = np.array(value)

    def cells(items):
        return np.array([item.name for item in items], dtype=object)
    data["Q_index"] = cells(system.configs)
    data["p_index"] = cells(system.dyn_configs)
    data["v_index"] = cells(system.kin_configs)
    data["u_index"] = cells(system.inputs)
    data["rho_index"] = cells(system.kin_configs)
    scipy.io.savemat(filename, data)


def load_trajectory(filename, system=None):
    """Read a file written by save_trajectory (system.py:1233-1304).  Without ``system``: the raw arrays with
    their name lists, ``(t, (Q_index, Q), (p_index, p), ...)``; with it: ``(t, Q, p, v, u, rho)`` re-ordered by
    name into the system's layout, unknown columns zero."""
    import scipy.io
    data = scipy.io.loadmat(filename)
    t = data["time"].squeeze()
    keys = ("Q", "p", "v", "u", "rho")
    arrays = dict((key, data.get(key, None)) for key in keys)
    names = dict((key, [str(c[0]).strip() for c in data[key + "_index"].ravel()]) for key in keys)
    if system is None:
        return (t,) + tuple((names[key], arrays[key]) for key in keys)
    layout = {"Q": (system.configs, len(t), lambda c: c.index), "p": (system.dyn_configs, len(t), lambda c: c.index),
              "v": (system.kin_configs, len(t), lambda c: c.k_index), "u": (system.inputs, len(t) - 1, lambda c: c.index),
              "rho": (system.kin_configs, len(t) - 1, lambda c: c.k_index)}
    out = []
    for key in keys:
        if arrays[key] is None:
            out.append(None)
            continue
        items, rows, col = layout[key]
        full = np.zeros((rows, len(items)))
        for item in items:
            if item.name in names[key]:
                full[:, col(item)] = arrays[key][:, names[key].index(item.name)]
        out.append(full)
    return (t,) + tuple(out)
