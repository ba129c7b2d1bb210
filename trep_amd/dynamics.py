"""Potentials, forces and holonomic constraints supported by the device path.

Host-side parameter holders only; the arithmetic lives in the HIP library.
Reference classes mirrored (constructor arguments, attribute names, ordering
side effects such as *when* a Distance constraint creates its kinematic length
config):

  Potential / Gravity       trep/potential.py:7-45,  trep/potentials/gravity.py:6-30
  Force / Damping / ConfigForce
                            trep/force.py:6-30, trep/forces/damping.py:8-58,
                            trep/forces/configforce.py:5-25
  Constraint / Distance / PointToPoint{1,2,3}D
                            trep/constraint.py:6-55, trep/constraints/distance.py:14-79,
                            trep/constraints/point.py:5-108

Types the reference has but no BASELINE config uses (springs, wrenches, linear
damper, point-on-plane, Python-defined callbacks) are out of scope.
"""
import numpy as np

from .config import Config, Input
from . import element_queries as _eq


class Potential(object):
    def __init__(self, system, name=None):
        self._system = system
        self.name = name
        system._add_potential(self)

    system = property(lambda self: self._system)

    # value / derivative queries at the system's current configuration (potential.py:42-76); host-side, see element_queries.py
    def _query(self, configs):
        raise NotImplementedError("%s defines no potential queries" % type(self).__name__)

    def V(self):
        return self._query(())

    def V_dq(self, q1):
        _eq.check_configs(q1)
        return self._query((q1,))

    def V_dqdq(self, q1, q2):
        _eq.check_configs(q1, q2)
        return self._query((q1, q2))

    def V_dqdqdq(self, q1, q2, q3):
        _eq.check_configs(q1, q2, q3)
        return self._query((q1, q2, q3))

    # finite-difference checks of the derivative queries (potential.py:78-110)
    def validate_V_dq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._system.test_derivative_dq(self.V, self.V_dq, delta, tolerance, verbose=verbose,
                                               test_name='%s.V_dq()' % type(self).__name__)

    def validate_V_dqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return all([self._system.test_derivative_dq(lambda q1=q1: self.V_dq(q1), lambda q2, q1=q1: self.V_dqdq(q1, q2), delta, tolerance,
                                                    verbose=verbose, test_name='%s.V_dqdq()' % type(self).__name__)
                    for q1 in self._system.configs])

    def validate_V_dqdqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return all([self._system.test_derivative_dq(lambda q1=q1, q2=q2: self.V_dqdq(q1, q2), lambda q3, q1=q1, q2=q2: self.V_dqdqdq(q1, q2, q3),
                                                    delta, tolerance, verbose=verbose, test_name='%s.V_dqdqdq()' % type(self).__name__)
                    for q1, q2 in _eq.pairs(self._system, 2)])


class Gravity(Potential):
    """V = -sum_masses m * g . p   (potentials/gravity.c:12-25)."""

    def __init__(self, system, gravity=(0.0, 0.0, -9.8), name=None):
        Potential.__init__(self, system, name)
        self.gravity = gravity

    def __repr__(self):
        return "<Gravity %f %f %f>" % tuple(self.gravity)

    @property
    def gravity(self):
        return np.array(self._gravity)

    @gravity.setter
    def gravity(self, g):
        self._gravity = (float(g[0]), float(g[1]), float(g[2]))
        self._system._structure_changed()

    def _query(self, configs):
        return _eq.gravity(self, configs)


class ConfigSpring(Potential):
    """V = 1/2 k (q - q0)^2 on one configuration variable (potentials/configspring.py:15-54)."""

    def __init__(self, system, config, k, q0=0.0, name=None):
        Potential.__init__(self, system, name)
        if not system.get_config(config):
            raise ValueError("Could not find config %r" % config)
        self._config = system.get_config(config)
        self._k = float(k)
        self._q0 = float(q0)
        system._structure_changed()

    def __repr__(self):
        return "<ConfigSpring %r k=%f q0=%f>" % (self._config.name, self._k, self._q0)

    config = property(lambda self: self._config)

    @property
    def k(self):
        return self._k

    @k.setter
    def k(self, value):
        self._k = float(value)
        self._system._structure_changed()

    @property
    def q0(self):
        return self._q0

    @q0.setter
    def q0(self, value):
        self._q0 = float(value)
        self._system._structure_changed()

    def _query(self, configs):
        return _eq.config_spring(self, configs)


class NonlinearConfigSpring(Potential):
    """A spring on one configuration variable whose force is a curve: dV/dq = -f(m q + b), f a ``trep_amd.Spline``
    (potentials/nonlinear_config_spring.py:15-40, _trep/potentials/nonlinear_config_spring.c:15-61).  As in the reference the
    potential VALUE is not defined (``V()`` returns 0: only its derivatives enter simulation and optimisation) and the
    spline is copied at construction."""

    def __init__(self, system, config, spline, m=1.0, b=0.0, name=None):
        from .spline import Spline
        Potential.__init__(self, system, name)
        if not system.get_config(config):
            raise ValueError("Could not find config %r" % config)
        self._config = system.get_config(config)
        if not isinstance(spline, Spline):
            raise TypeError("spline must be a trep_amd.Spline")
        self._spline = spline.copy()
        self._m = float(m)
        self._b = float(b)
        system._structure_changed()

    def __repr__(self):
        return "<NonlinearConfigSpring %r m=%f b=%f>" % (self._config.name, self._m, self._b)

    config = property(lambda self: self._config)
    spline = property(lambda self: self._spline)
    m = property(lambda self: self._m)
    b = property(lambda self: self._b)

    def _query(self, configs):
        return _eq.nonlinear_config_spring(self, configs)


class LinearSpring(Potential):
    """V = 1/2 k (|p(frame1) - p(frame2)| - x0)^2 (potentials/linearspring.py:15-70)."""

    def __init__(self, system, frame1, frame2, k, x0=0, name=None):
        Potential.__init__(self, system, name)
        if not system.get_frame(frame1):
            raise ValueError("Could not find frame %r" % frame1)
        self._frame1 = system.get_frame(frame1)
        if not system.get_frame(frame2):
            raise ValueError("Could not find frame %r" % frame2)
        self._frame2 = system.get_frame(frame2)
        self._k = float(k)
        self._x0 = float(x0)
        system._structure_changed()

    def __repr__(self):
        return "<LinearSpring %r %r k=%f x0=%f>" % (self._frame1.name, self._frame2.name, self._k, self._x0)

    frame1 = property(lambda self: self._frame1)
    frame2 = property(lambda self: self._frame2)

    @property
    def k(self):
        return self._k

    @k.setter
    def k(self, value):
        self._k = float(value)
        self._system._structure_changed()

    @property
    def x0(self):
        return self._x0

    @x0.setter
    def x0(self, value):
        self._x0 = float(value)
        self._system._structure_changed()

    def _query(self, configs):
        return _eq.linear_spring(self, configs)


class Force(object):
    def __init__(self, system, name=None):
        self._system = system
        self.name = name
        system._add_force(self)

    system = property(lambda self: self._system)

    def _create_input(self, name=None):
        new_input = Input(self._system, name)
        new_input._force = self
        return new_input

    # generalized force on config q and its derivatives at the system's current state (force.py:46-145); host-side, see
    # element_queries.py.  The defaults (zero) are what the reference's elements return for the derivatives they do not have.
    def f(self, q):
        return 0.0

    def f_dq(self, q, q1):
        return 0.0

    def f_ddq(self, q, dq1):
        return 0.0

    def f_du(self, q, u1):
        return 0.0

    def f_dqdq(self, q, q1, q2):
        return 0.0

    def f_ddqdq(self, q, dq1, q2):
        return 0.0

    def f_ddqddq(self, q, dq1, dq2):
        return 0.0

    def f_dudq(self, q, u1, q2):
        return 0.0

    def f_duddq(self, q, u1, dq2):
        return 0.0

    def f_dudu(self, q, u1, u2):
        return 0.0

    def validate_f_dq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        """f_dq() against central differences of f() (force.py:147-160)."""
        return all([self._system.test_derivative_dq(lambda q=q: self.f(q), lambda q1, q=q: self.f_dq(q, q1), delta, tolerance,
                                                    verbose=verbose, test_name='%s.f_dq()' % type(self).__name__)
                    for q in self._system.configs])


class Damping(Force):
    """f_i = -c_i * dq_i on every dynamic config (forces/damping.c:13-27)."""

    def __init__(self, system, default=0.0, coefficients={}, name=None):
        Force.__init__(self, system, name)
        self._default = float(default)
        self.coefficients = {}
        for config, coeff in coefficients.items():
            self.coefficients[system.get_config(config)] = float(coeff)

    def coefficient_array(self):
        """One coefficient per dynamic config, in system.dyn_configs order."""
        out = np.ones(self._system.nQd, dtype=np.float64) * self._default
        for config, coeff in self.coefficients.items():
            out[config.index] = coeff
        return out

    def get_damping_coefficient(self, config):
        config = self._system.get_config(config)
        if config is None:
            raise ValueError("Couldn't find config")
        return self.coefficients.get(config, self._default)

    def set_damping_coefficient(self, config, coeff):
        if coeff is None:
            self.coefficients.pop(self._system.get_config(config), None)
        else:
            self.coefficients[self._system.get_config(config)] = float(coeff)
        self._system._structure_changed()

    def f(self, q):
        return -self.get_damping_coefficient(q) * q.dq if not q.kinematic else 0.0

    def f_ddq(self, q, dq1):
        return -self.get_damping_coefficient(q) if (q is dq1 and not q.kinematic) else 0.0

    @property
    def default(self):
        return self._default

    @default.setter
    def default(self, value):
        self._default = float(value)
        self._system._structure_changed()


class ConfigForce(Force):
    """Generalized force u applied directly to one config (forces/configforce.c:13-37)."""

    def __init__(self, system, config, finput, name=None):
        Force.__init__(self, system, name)
        if not system.get_config(config):
            raise ValueError("Could not find config %r" % config)
        self._config = system.get_config(config)
        self._input = self._create_input(finput)

    finput = property(lambda self: self._input)
    config = property(lambda self: self._config)

    def f(self, q):
        return self._input.u if q is self._config else 0.0

    def f_du(self, q, u1):
        return 1.0 if (q is self._config and u1 is self._input) else 0.0


class HybridWrench(Force):
    """A wrench applied at a frame, force in world coordinates and torque in body coordinates; every component is a
    constant or a new input (forces/hybridwrench.py:15-33).  The device path implements the force part."""

    def __init__(self, system, frame, wrench=tuple(), name=None):
        Force.__init__(self, system, name)
        if not system.get_frame(frame):
            raise ValueError("Could not find frame %r" % frame)
        self._frame = system.get_frame(frame)
        wrench = (list(wrench) + [0.0] * 6)[:6]
        self._wrench_vars = [None] * 6
        self._wrench_cons = [0.0] * 6
        for i in range(6):
            if isinstance(wrench[i], str):
                self._wrench_vars[i] = self._create_input(wrench[i])
            else:
                self._wrench_cons[i] = float(wrench[i])
        system._structure_changed()

    frame = property(lambda self: self._frame)

    @property
    def wrench_val(self):
        return [v.u if v is not None else c for (v, c) in zip(self._wrench_vars, self._wrench_cons)]

    @wrench_val.setter
    def wrench_val(self, wrench):
        for i, value in enumerate(wrench[:6]):
            if self._wrench_vars[i] is not None:
                self._wrench_vars[i].u = value
            else:
                self._wrench_cons[i] = float(value)
        self._system._structure_changed()


    _twist_kind = "hybrid"

    def _twist(self, q, configs):
        return _eq.wrench_twist(self, q, configs, self._twist_kind)

    def f(self, q):
        return float(self._twist(q, ()).dot(_eq.wrench_value(self)))

    def f_dq(self, q, q1):
        return float(self._twist(q, (q1,)).dot(_eq.wrench_value(self)))

    def f_dqdq(self, q, q1, q2):
        return float(self._twist(q, (q1, q2)).dot(_eq.wrench_value(self)))

    def f_du(self, q, u1):
        vec = self._twist(q, ())
        return float(sum(vec[i] for i in range(6) if self._wrench_vars[i] is u1))

    def f_dudq(self, q, u1, q2):
        vec = self._twist(q, (q2,))
        return float(sum(vec[i] for i in range(6) if self._wrench_vars[i] is u1))


class SpatialWrench(HybridWrench):
    """A wrench given in spatial (world) coordinates: its six components multiply the spatial twist of each joint of the
    frame's path, i.e. the force acts at the point of the frame that coincides with the world origin
    (forces/spatialwrench.py:14-37, spatialwrench.c:16-38)."""
    _twist_kind = "spatial"


class BodyWrench(HybridWrench):
    """A wrench given in the coordinates of the frame it is applied to: its six components multiply the body twist of each
    joint of the frame's path (forces/bodywrench.py, bodywrench.c:16-38)."""
    _twist_kind = "body"


class LinearDamper(Force):
    """A viscous damper between the origins of two frames: force -c d|p1 - p2|/dt along the line between them
    (forces/lineardamper.py:14-60, lineardamper.c:12-45)."""

    def __init__(self, system, frame1, frame2, c, name=None):
        Force.__init__(self, system, name)
        if not system.get_frame(frame1):
            raise ValueError("Could not find frame %r" % frame1)
        self._frame1 = system.get_frame(frame1)
        if not system.get_frame(frame2):
            raise ValueError("Could not find frame %r" % frame2)
        self._frame2 = system.get_frame(frame2)
        self._c = float(c)
        system._structure_changed()

    frame1 = property(lambda self: self._frame1)
    frame2 = property(lambda self: self._frame2)

    @property
    def c(self):
        return self._c

    @c.setter
    def c(self, value):
        self._c = float(value)
        self._system._structure_changed()

    # lineardamper.c:14-92 with x = |p1 - p2| and v = dx/dt = sum_k x_k dq_k
    def _on(self, *configs):
        return all(self._frame1.uses_config(c) or self._frame2.uses_config(c) for c in configs)

    def f(self, q):
        return -self._c * _eq.velocity(self) * _eq.length_dq(self, q) if self._on(q) else 0.0

    def f_dq(self, q, q1):
        if not self._on(q, q1):
            return 0.0
        return -self._c * (_eq.velocity_dq(self, q1) * _eq.length_dq(self, q) + _eq.velocity(self) * _eq.length_dqdq(self, q, q1))

    def f_ddq(self, q, dq1):
        return -self._c * _eq.length_dq(self, dq1) * _eq.length_dq(self, q) if self._on(q, dq1) else 0.0

    def f_dqdq(self, q, q1, q2):
        if not self._on(q, q1, q2):
            return 0.0
        return -self._c * (_eq.velocity_dqdq(self, q1, q2) * _eq.length_dq(self, q) + _eq.velocity_dq(self, q1) * _eq.length_dqdq(self, q, q2) +
                           _eq.velocity_dq(self, q2) * _eq.length_dqdq(self, q, q1) + _eq.velocity(self) * _eq.length_dqdqdq(self, q, q1, q2))

    def f_ddqdq(self, q, dq1, q2):
        """(The second term uses d x / d q2 where d2 x / dq dq2 is meant -- the reference's expression, lineardamper.c:88.)"""
        if not self._on(q, dq1, q2):
            return 0.0
        return -self._c * (_eq.length_dqdq(self, dq1, q2) * _eq.length_dq(self, q) + _eq.length_dq(self, dq1) * _eq.length_dq(self, q2))


class Constraint(object):
    def __init__(self, system, name=None, tolerance=1e-10):
        self._system = system
        self.name = name
        self.tolerance = tolerance
        self._index = -1
        system._add_constraint(self)

    system = property(lambda self: self._system)

    @property
    def index(self):
        self._system._sync()
        return self._index

    def get_actual_distance(self):
        p1 = self.frame1.p()
        p2 = self.frame2.p()
        return ((p1[0] - p2[0]) ** 2.0 + (p1[1] - p2[1]) ** 2.0 + (p1[2] - p2[2]) ** 2.0) ** 0.5

    # value / derivative queries at the system's current configuration (constraint.py:56-102); host-side, see element_queries.py
    def _query(self, configs):
        raise NotImplementedError("%s defines no constraint queries" % type(self).__name__)

    def h(self):
        return self._query(())

    def h_dq(self, q1):
        _eq.check_configs(q1)
        return self._query((q1,))

    def h_dqdq(self, q1, q2):
        _eq.check_configs(q1, q2)
        return self._query((q1, q2))

    def h_dqdqdq(self, q1, q2, q3):
        _eq.check_configs(q1, q2, q3)
        return self._query((q1, q2, q3))

    def h_dqdqdqdq(self, q1, q2, q3, q4):
        _eq.check_configs(q1, q2, q3, q4)
        return self._query((q1, q2, q3, q4))

    # finite-difference checks (constraint.py:104-150)
    def validate_h_dq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._system.test_derivative_dq(self.h, self.h_dq, delta, tolerance, verbose=verbose, test_name='%s.h_dq()' % type(self).__name__)

    def _validate_higher(self, order, delta, tolerance, verbose):
        lower = (self.h_dq, self.h_dqdq, self.h_dqdqdq)[order - 2]
        upper = (self.h_dqdq, self.h_dqdqdq, self.h_dqdqdqdq)[order - 2]
        return all([self._system.test_derivative_dq(lambda qs=qs: lower(*qs), lambda qn, qs=qs: upper(*(qs + (qn,))), delta, tolerance,
                                                    verbose=verbose, test_name='%s.h_%s()' % (type(self).__name__, 'dq' * order))
                    for qs in _eq.pairs(self._system, order - 1)])

    def validate_h_dqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._validate_higher(2, delta, tolerance, verbose)

    def validate_h_dqdqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._validate_higher(3, delta, tolerance, verbose)

    def validate_h_dqdqdqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._validate_higher(4, delta, tolerance, verbose)


class Distance(Constraint):
    """h = |p1 - p2|^2 - d^2, d constant or a new kinematic config (constraints/distance.c:16-33)."""

    def __init__(self, system, frame1, frame2, distance, name=None):
        Constraint.__init__(self, system, name)
        assert frame1 is not None
        assert frame2 is not None
        self._frame1 = system.get_frame(frame1)
        self._frame2 = system.get_frame(frame2)
        if isinstance(distance, str):
            self._config = Config(system, name=distance, kinematic=True)
            self._distance = 0.0
        else:
            self._config = None
            self._distance = float(distance)

    def __repr__(self):
        mid = "'%s'" % self._config.name if self._config else "%f" % self._distance
        return "<DistanceConstraint '%s' %s '%s'>" % (self.frame1.name, mid, self.frame2.name)

    config = property(lambda self: self._config)
    frame1 = property(lambda self: self._frame1)
    frame2 = property(lambda self: self._frame2)

    def _query(self, configs):
        return _eq.distance(self, configs)

    @property
    def distance(self):
        return self._config.q if self._config else self._distance

    @distance.setter
    def distance(self, value):
        if self._config:
            self._config.q = value
        else:
            self._distance = float(value)
            self._system._structure_changed()


class PointToPoint1D(Constraint):
    """h = (p1 - p2)[axis]   (constraints/point.c:16-27)."""

    _AXES = {"x": 0, "y": 1, "z": 2, "X": 0, "Y": 1, "Z": 2}

    def __init__(self, system, axis, frame1, frame2, name=None):
        Constraint.__init__(self, system, name)
        self._frame1 = system.get_frame(frame1)
        self._frame2 = system.get_frame(frame2)
        self.axis = axis
        self._component = self._AXES[axis]

    def __repr__(self):
        return "<PointToPointConstraint %s-axis '%s' '%s'>" % (self.axis, self.frame1.name, self.frame2.name)

    frame1 = property(lambda self: self._frame1)
    frame2 = property(lambda self: self._frame2)
    component = property(lambda self: self._component)

    def _query(self, configs):
        return _eq.point_1d(self, configs)


class PointOnPlane(Constraint):
    """h = (R_plane n) . (p_plane - p_point): a point held in a plane that moves with another frame
    (constraints/plane.py:7-37, plane.c:13-26)."""

    def __init__(self, system, plane_frame, plane_normal, point_frame, name=None):
        Constraint.__init__(self, system, name)
        self._plane_frame = system.get_frame(plane_frame)
        self._point_frame = system.get_frame(point_frame)
        self.normal = plane_normal

    def __repr__(self):
        return "<PointOnPlane plane_frame='%s' normal=(%f %f %f) point_frame='%s'>" % (
            (self._plane_frame.name,) + tuple(self._normal) + (self._point_frame.name,))

    plane_frame = property(lambda self: self._plane_frame)
    point_frame = property(lambda self: self._point_frame)
    frame1 = property(lambda self: self._plane_frame)      # descriptor order: frame1 = plane, frame2 = point
    frame2 = property(lambda self: self._point_frame)

    @property
    def normal(self):
        return np.array(self._normal)

    @normal.setter
    def normal(self, normal):
        self._normal = (float(normal[0]), float(normal[1]), float(normal[2]))
        self._system._structure_changed()

    def _query(self, configs):
        return _eq.point_on_plane(self, configs)


class _PointGroup(object):
    """PointToPoint2D/3D are not constraints themselves: they add 1-D ones."""

    def __init__(self, system, axes, frame1, frame2, name):
        assert frame1 is not None
        assert frame2 is not None
        self.frame1 = system.get_frame(frame1)
        self.frame2 = system.get_frame(frame2)
        for axis in axes:
            PointToPoint1D(system, axis, frame1, frame2, name)

    get_actual_distance = Constraint.get_actual_distance


class PointToPoint3D(_PointGroup):
    def __init__(self, system, frame1, frame2, name=None):
        _PointGroup.__init__(self, system, "xyz", frame1, frame2, name)


class PointToPoint2D(_PointGroup):
    _PLANES = {"yz": "yz", "zy": "yz", "xz": "xz", "zx": "xz", "xy": "xy", "yx": "xy"}

    def __init__(self, system, plane, frame1, frame2, name=None):
        _PointGroup.__init__(self, system, self._PLANES[plane.lower()], frame1, frame2, name)
