"""Per-element value / derivative queries of potentials, forces and constraints at the system's current state
(reference accessors: potential.py:42-76, force.py:46-145, constraint.py:56-102 and the C implementations they call).

Host-side numpy helpers for inspection, validation (``validate_*``) and tests, built on the frame queries of
``trep_amd.frame`` (``p_dq`` ..., ``g_dq`` ..., ``g_inv_dq`` ...).  The integrator never calls them: its potentials, forces
and constraints are evaluated on the device from the flattened tables (csrc/mvi_core.hpp).  Each function restates the
formula of the reference element it cites; quirks of the reference that its own tests see are reproduced and named.
"""
import itertools

import numpy as np


def _p(frame, configs):
    """d^n p / d(configs) of a frame origin as a 3-vector (zero if the frame does not depend on one of them)."""
    n = len(configs)
    if n == 0:
        return frame.p()[:3]
    return (frame.p_dq, frame.p_dqdq, frame.p_dqdqdq, frame.p_dqdqdqdq)[n - 1](*configs)[:3]


def _sep(el, configs):
    """d^n (p(frame1) - p(frame2)) / d(configs)."""
    return _p(el.frame1, configs) - _p(el.frame2, configs)


def _uses(el, configs):
    return all(el.frame1.uses_config(q) or el.frame2.uses_config(q) for q in configs)


# ---- distance between two frame origins and its rate (the reference's TapeMeasure restricted to two frames,
#      tapemeasure.c; used by LinearSpring and LinearDamper) ------------------------------------------------------
def length(el):
    v = _sep(el, ())
    return float(np.sqrt(v.dot(v)))


def length_dq(el, a):
    v = _sep(el, ())
    return float(v.dot(_sep(el, (a,))) / np.sqrt(v.dot(v)))


def length_dqdq(el, a, b):
    v, va, vb, vab = _sep(el, ()), _sep(el, (a,)), _sep(el, (b,)), _sep(el, (a, b))
    x = np.sqrt(v.dot(v))
    xa, xb = v.dot(va) / x, v.dot(vb) / x
    return float((va.dot(vb) + v.dot(vab) - xa * xb) / x)


def length_dqdqdq(el, a, b, c):
    v = _sep(el, ())
    va, vb, vc = _sep(el, (a,)), _sep(el, (b,)), _sep(el, (c,))
    vab, vac, vbc, vabc = _sep(el, (a, b)), _sep(el, (a, c)), _sep(el, (b, c)), _sep(el, (a, b, c))
    x = np.sqrt(v.dot(v))
    xa, xb, xc = v.dot(va) / x, v.dot(vb) / x, v.dot(vc) / x
    xab = (va.dot(vb) + v.dot(vab) - xa * xb) / x
    xac = (va.dot(vc) + v.dot(vac) - xa * xc) / x
    xbc = (vb.dot(vc) + v.dot(vbc) - xb * xc) / x
    return float((vac.dot(vb) + va.dot(vbc) + vc.dot(vab) + v.dot(vabc) - xac * xb - xa * xbc - xab * xc) / x)


def _drivers(el):
    return [q for q in el.system.configs if el.frame1.uses_config(q) or el.frame2.uses_config(q)]


def velocity(el):
    return sum(length_dq(el, k) * k.dq for k in _drivers(el))


def velocity_dq(el, a):
    return sum(length_dqdq(el, k, a) * k.dq for k in _drivers(el))


def velocity_dqdq(el, a, b):
    return sum(length_dqdqdq(el, k, a, b) * k.dq for k in _drivers(el))


# ---- potentials ------------------------------------------------------------------------------------------------
def gravity(pot, configs):
    """V = -sum m g . p and its config derivatives (gravity.c:12-94)."""
    g = np.asarray(pot.gravity, dtype=float)
    return float(-sum(f.mass * g.dot(_p(f, configs)) for f in pot.system.masses))


def config_spring(pot, configs):
    """1/2 k (q - q0)^2: k (q - q0), k, 0 when every argument is the spring's config (configspring.c:15-45)."""
    n = len(configs)
    if n == 0:
        return 0.5 * pot.k * (pot.config.q - pot.q0) ** 2
    if any(q is not pot.config for q in configs):
        return 0.0
    return (pot.k * (pot.config.q - pot.q0), pot.k, 0.0)[n - 1]


def nonlinear_config_spring(pot, configs):
    """dV/dq = -y(m q + b); the value itself is not defined (0) and the third derivative carries the reference's sign,
    -y'' * -m * m (nonlinear_config_spring.c:15-61)."""
    n = len(configs)
    if n == 0 or any(q is not pot.config for q in configs):
        return 0.0
    x = pot.m * pot.config.q + pot.b
    return (-pot.spline.y(x), -pot.spline.dy(x) * pot.m, -pot.spline.ddy(x) * -pot.m * pot.m)[n - 1]


def linear_spring(pot, configs):
    """1/2 k (x - x0)^2 with x the distance of the two frame origins; value, gradient and Hessian only, as in the
    reference (linearspring.c:16-88)."""
    n = len(configs)
    x = length(pot)
    if n == 0:
        return 0.5 * pot.k * (x - pot.x0) ** 2
    if not _uses(pot, configs):
        return 0.0
    if n == 1:
        return pot.k * (x - pot.x0) * length_dq(pot, configs[0])
    if n == 2:
        a, b = configs
        return pot.k * length_dq(pot, a) * length_dq(pot, b) + pot.k * (x - pot.x0) * length_dqdq(pot, a, b)
    raise NotImplementedError("LinearSpring defines no V_dqdqdq (linearspring.c:86-88)")


# ---- constraints -----------------------------------------------------------------------------------------------
def distance(con, configs):
    """h = |p1 - p2|^2 - d^2 (distance.c:16-133): product rule over the separation vector; a length config enters
    the first two derivatives only."""
    n = len(configs)
    total = 0.0
    # sum over the ways to split the differentiation variables between the two factors of v . v
    for mask in range(1 << n):
        left = tuple(configs[i] for i in range(n) if mask >> i & 1)
        right = tuple(configs[i] for i in range(n) if not mask >> i & 1)
        total += _sep(con, left).dot(_sep(con, right))
    if n == 0:
        total -= con.distance ** 2
    elif con.config is not None and all(q is con.config for q in configs):
        total -= (2.0 * con.distance, 2.0)[n - 1] if n <= 2 else 0.0
    return float(total)


def point_1d(con, configs):
    """h = (p1 - p2)[axis] (point.c:16-75)."""
    return float(_sep(con, configs)[con.component])


def _g(frame, configs):
    n = len(configs)
    if n == 0:
        return frame.g()
    return (frame.g_dq, frame.g_dqdq, frame.g_dqdqdq, frame.g_dqdqdqdq)[n - 1](*configs)


def point_on_plane(con, configs):
    """h = (R n) . (p_plane - p_point), R the rotation of the plane frame (plane.c:13-110): product rule over the two
    factors."""
    nrm = np.asarray(con.normal, dtype=float)
    n = len(configs)
    total = 0.0
    for mask in range(1 << n):
        left = tuple(configs[i] for i in range(n) if mask >> i & 1)
        right = tuple(configs[i] for i in range(n) if not mask >> i & 1)
        total += _g(con.plane_frame, left)[:3, :3].dot(nrm).dot(_sep(con, right))
    return float(total)


# ---- forces ----------------------------------------------------------------------------------------------------
def _unhat(m):
    return np.array([m[0, 3], m[1, 3], m[2, 3], m[2, 1], m[0, 2], m[1, 0]])


def wrench_twist(force, q, configs, kind):
    """The 6-vector that multiplies the wrench components in f(q), differentiated with respect to `configs`:
    hybrid (hybridwrench.c:15-110): the linear part of g_dq(q) and the angular part of g_dq(q) g^-1;
    spatial (spatialwrench.c:16-100): all of g_dq(q) g^-1; body (bodywrench.c:16-100): all of g^-1 g_dq(q)."""
    frame = force.frame
    if not all(frame.uses_config(c) for c in (q,) + tuple(configs)):
        return np.zeros(6)
    n = len(configs)
    prod = np.zeros((4, 4))
    for mask in range(1 << n):
        with_g = tuple(configs[i] for i in range(n) if mask >> i & 1)
        with_inv = tuple(configs[i] for i in range(n) if not mask >> i & 1)
        gd, gi = frame._g_n((q,) + with_g), frame._g_inv_n(with_inv)
        prod += gi.dot(gd) if kind == "body" else gd.dot(gi)
    vec = _unhat(prod)
    if kind == "hybrid":
        vec[:3] = _unhat(frame._g_n((q,) + tuple(configs)))[:3]
    return vec


def wrench_value(force):
    return np.array([v.u if v is not None else c for v, c in zip(force._wrench_vars, force._wrench_cons)])


def check_configs(*configs):
    from .config import Config
    for q in configs:
        if not isinstance(q, Config):
            raise TypeError("expected a Config, got %r" % (q,))


def pairs(system, repeat):
    return itertools.product(system.configs, repeat=repeat)
