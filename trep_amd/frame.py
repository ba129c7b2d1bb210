"""Coordinate frames: the tree that defines a mechanical system's kinematics.

Host-side mirror of the reference's ``trep.Frame`` and the ``tx/ty/.../const_se3``
frame-definition helpers (/root/reference/trep/frame.py:11-36, 112-162, 195-220,
321-358, 658-691).  A frame is a single one-parameter SE(3) transform from its
parent (translation along / rotation about one axis, driven by a ``Config`` or
by a constant), or a constant SE(3).  Model building and plain numpy kinematic
queries of a single frame (``g()``, ``p()``, ``vb()`` and their derivatives, for set-up,
inspection and tests) live here; the integrator and all of its derivatives run on the device.
"""
import math

import numpy as np

from .config import Config


class FrameTransform(object):
    """Transform-kind tag; ``code`` is the integer stored in the flattened tables."""

    def __init__(self, name, code):
        self.name = name
        self.code = code

    def __repr__(self):
        return self.name


WORLD = FrameTransform("WORLD", 0)
TX = FrameTransform("TX", 1)
TY = FrameTransform("TY", 2)
TZ = FrameTransform("TZ", 3)
RX = FrameTransform("RX", 4)
RY = FrameTransform("RY", 5)
RZ = FrameTransform("RZ", 6)
CONST_SE3 = FrameTransform("CONST_SE3", 7)

_PARAMETRIC = (TX, TY, TZ, RX, RY, RZ)


class FrameDef(object):
    """Entry of the nested-list tree description accepted by ``import_frames``."""

    def __init__(self, transform_type, param, name, kinematic, mass):
        self.transform_type = transform_type
        self.param = param
        self.name = name
        self.kinematic = kinematic
        self.mass = mass

    def __repr__(self):
        return "<FrameDef %s>" % (self.transform_type,)


def _definer(kind):
    def define(param, name=None, kinematic=False, mass=0.0):
        return FrameDef(kind, param, name, kinematic, mass)
    define.__name__ = kind.name.lower()
    return define


tx = _definer(TX)
ty = _definer(TY)
tz = _definer(TZ)
rx = _definer(RX)
ry = _definer(RY)
rz = _definer(RZ)


def const_se3(se3, name=None, kinematic=False, mass=0.0):
    return FrameDef(CONST_SE3, se3, name, kinematic, mass)


def const_txyz(xyz, name=None, kinematic=False, mass=0.0):
    return FrameDef(CONST_SE3, [[1, 0, 0], [0, 1, 0], [0, 0, 1], xyz], name, kinematic, mass)


def local_transform(kind, x, const_lg=None):
    """4x4 local transform of one frame at parameter value x (frame.c:839-1068)."""
    m = np.eye(4)
    if kind is WORLD:
        return m
    if kind is CONST_SE3:
        return np.array(const_lg, dtype=float)
    if kind is TX:
        m[0, 3] = x
    elif kind is TY:
        m[1, 3] = x
    elif kind is TZ:
        m[2, 3] = x
    else:
        c, s = math.cos(x), math.sin(x)
        if kind is RX:
            m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
        elif kind is RY:
            m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
        elif kind is RZ:
            m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


class Frame(object):
    def __init__(self, parent, transform, param, name=None, kinematic=False, mass=0.0):
        self.name = name
        self._config = None
        self._value = 0.0
        self._lg_const = np.eye(4)
        self._children = tuple()
        self._cache_index = tuple()
        self._cache_size = 0
        self._mass = self._Ixx = self._Iyy = self._Izz = 0.0
        self._transform = transform
        if transform is WORLD:
            self._system = parent
            self._parent = None
        elif transform in _PARAMETRIC or transform is CONST_SE3:
            if not isinstance(parent, Frame):
                raise TypeError("parent must be a Frame")
            self._system = parent.system
            self._parent = parent
            parent._children += (self,)
            if transform is CONST_SE3:
                self.set_SE3(param[0], param[1], param[2], param[3])
            elif isinstance(param, str):
                self._config = Config(self._system, param, kinematic=kinematic)
            else:
                self._value = float(param)
        else:
            raise Exception("Unknown frame transform: %r" % (transform,))
        self.set_mass(mass)

    def __repr__(self):
        if self._transform is WORLD:
            return "<Frame '%s'>" % self.name
        if self._config is None:
            return "<Frame '%s' %s(%s) %f>" % (self.name, self._transform, self.transform_value, self.mass)
        return "<Frame '%s' %s(%s)>" % (self.name, self._transform, self._config.name)

    # -- tree ---------------------------------------------------------------
    system = property(lambda self: self._system)
    config = property(lambda self: self._config)
    parent = property(lambda self: self._parent)
    children = property(lambda self: self._children)
    transform_type = property(lambda self: self._transform)

    def tree_view(self, indent=0):
        lines = [indent * " " + repr(self)]
        lines += [c.tree_view(indent + 3) for c in self._children]
        return "\n".join(lines)

    def flatten_tree(self):
        """Depth-first, parent before children, children in insertion order."""
        out = []
        stack = [self]
        while stack:
            f = stack.pop()
            out.append(f)
            stack.extend(reversed(f._children))
        return out

    def import_frames(self, children):
        """Build a sub-tree from the nested [def, [children...], def, ...] list form."""
        self.system.hold_structure_changes()
        try:
            pending = list(children)
            i = 0
            while i < len(pending):
                info = pending[i]
                if not isinstance(info, FrameDef):
                    raise TypeError("Frame definition expected instead of: %r" % (info,))
                frame = Frame(self, info.transform_type, info.param, name=info.name,
                              kinematic=info.kinematic, mass=info.mass)
                i += 1
                if i < len(pending) and isinstance(pending[i], list):
                    frame.import_frames(pending[i])
                    i += 1
        finally:
            self.system.resume_structure_changes()

    def uses_config(self, q):
        self._system._sync()
        return self._cache_index[q._config_gen] is q

    @property
    def cache_index(self):
        self._system._sync()
        return self._cache_index

    @property
    def cache_size(self):
        self._system._sync()
        return self._cache_size

    # -- parameters ---------------------------------------------------------
    @property
    def transform_value(self):
        return self._config.q if self._config is not None else self._value

    @transform_value.setter
    def transform_value(self, value):
        if self._config is not None:
            self._config.q = value
        else:
            self._value = float(value)
            self._system._structure_changed()

    def set_SE3(self, Rx=(1, 0, 0), Ry=(0, 1, 0), Rz=(0, 0, 1), p=(0, 0, 0)):
        """Orthonormalise (Rx, Ry) into a rotation and store the constant transform."""
        ax = np.array(Rx, dtype=float)
        ax = ax / np.linalg.norm(ax)
        az = np.cross(ax, np.array(Ry, dtype=float))
        az = az / np.linalg.norm(az)
        ay = np.cross(az, ax)
        ay = ay / np.linalg.norm(ay)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2] = ax[:3], ay[:3], az[:3]
        m[:3, 3] = [float(p[0]), float(p[1]), float(p[2])]
        self._lg_const = m
        self._system._structure_changed()

    def set_mass(self, mass, Ixx=0.0, Iyy=0.0, Izz=0.0):
        try:
            Ixx, Iyy, Izz, mass = mass[1], mass[2], mass[3], mass[0]
        except TypeError:
            pass
        self._mass, self._Ixx, self._Iyy, self._Izz = float(mass), float(Ixx), float(Iyy), float(Izz)
        self._system._structure_changed()

    def _inertia_prop(attr):
        def getter(self):
            return getattr(self, attr)

        def setter(self, v):
            setattr(self, attr, float(v))
            self._system._structure_changed()
        return property(getter, setter)

    mass = _inertia_prop("_mass")
    Ixx = _inertia_prop("_Ixx")
    Iyy = _inertia_prop("_Iyy")
    Izz = _inertia_prop("_Izz")
    del _inertia_prop

    # -- host forward kinematics (setup only) --------------------------------
    def export_frames(self, tabs=0, tab_size=4):
        """This frame and its children as source text in the nested-list form that import_frames reads
        (frame.py:223-262 of the reference)."""
        names = {TX: 'tx', TY: 'ty', TZ: 'tz', RX: 'rx', RY: 'ry', RZ: 'rz', CONST_SE3: 'const_se3'}
        params = []
        if self._transform is CONST_SE3:
            mat = self.lg()
            cols = ['[' + ', '.join('%s' % x for x in mat[:3, j]) + ']' for j in range(4)]
            params.append('[%s, %s, %s, %s]' % tuple(cols))
        elif self.config is None:
            params.append('%s' % self.transform_value)
        else:
            params.append("'%s'" % self.config.name)
            if self.config.kinematic:
                params.append('kinematic=True')
        if self.name:
            params.append("name='%s'" % self.name)
        if self.mass:
            if self.Ixx or self.Iyy or self.Izz:
                params.append('mass=[%s, %s, %s, %s]' % (self.mass, self.Ixx, self.Iyy, self.Izz))
            else:
                params.append('mass=%s' % self.mass)
        txt = ' ' * tab_size * tabs + '%s(%s)' % (names[self._transform], ', '.join(params))
        if self._children:
            txt += ', [\n' + ',\n'.join(c.export_frames(tabs + 1, tab_size) for c in self._children)
            txt += '\n' + ' ' * (tab_size * tabs + tab_size) + ']'
        return txt

    def lg(self):
        return local_transform(self._transform, self.transform_value, self._lg_const)

    # ---- kinematic queries of one frame (reference accessors frame.py:398-646) -------------------------------------
    # Host-side numpy helpers for inspection and tests; the integrator never calls them (its kinematics run on the
    # device in twist form, csrc/mvi_core.hpp).  Every parametric local transform is exp(T q) for a constant unit
    # twist T (frame.c:839-1068), so d^n lg / dq^n = lg T^n and d^n lg^-1 / dq^n = (-T)^n lg^-1; a derivative of the
    # world pose g = lg_1 lg_2 ... with respect to a set of configs puts T^(multiplicity) behind the frames they drive.
    def twist_hat(self):
        """Unit twist of the frame's local transform as a 4x4 se(3) matrix (frame.py:469)."""
        t = np.zeros((4, 4))
        k = self._transform
        if k is TX:
            t[0, 3] = 1.0
        elif k is TY:
            t[1, 3] = 1.0
        elif k is TZ:
            t[2, 3] = 1.0
        elif k is RX:
            t[2, 1], t[1, 2] = 1.0, -1.0
        elif k is RY:
            t[0, 2], t[2, 0] = 1.0, -1.0
        elif k is RZ:
            t[1, 0], t[0, 1] = 1.0, -1.0
        return t

    def _lg_n(self, n):
        return self.lg().dot(np.linalg.matrix_power(self.twist_hat(), n))

    def lg_dq(self):
        return self._lg_n(1)

    def lg_dqdq(self):
        return self._lg_n(2)

    def lg_dqdqdq(self):
        return self._lg_n(3)

    def lg_dqdqdqdq(self):
        return self._lg_n(4)

    def lg_inv(self):
        m = self.lg()
        out = np.eye(4)
        out[:3, :3] = m[:3, :3].T
        out[:3, 3] = -m[:3, :3].T.dot(m[:3, 3])
        return out

    def _lg_inv_n(self, n):
        return np.linalg.matrix_power(-self.twist_hat(), n).dot(self.lg_inv())

    def lg_inv_dq(self):
        return self._lg_inv_n(1)

    def lg_inv_dqdq(self):
        return self._lg_inv_n(2)

    def lg_inv_dqdqdq(self):
        return self._lg_inv_n(3)

    def lg_inv_dqdqdqdq(self):
        return self._lg_inv_n(4)

    def _path(self):
        chain = []
        f = self
        while f is not None:
            chain.append(f)
            f = f._parent
        chain.reverse()
        return chain

    def _check(self, configs):
        """True if the frame depends on every config of the list (else the derivative is zero, frame.py:56-75)."""
        for q in configs:
            if not isinstance(q, Config):
                raise TypeError("expected a Config, got %r" % (q,))
            if not self.uses_config(q):
                return False
        return True

    def _g_n(self, configs):
        m = np.eye(4)
        for f in self._path():
            m = m.dot(f.lg())
            n = sum(1 for q in configs if q is f._config) if f._config is not None else 0
            if n:
                m = m.dot(np.linalg.matrix_power(f.twist_hat(), n))
        return m

    def _g_inv_n(self, configs):
        m = np.eye(4)
        for f in reversed(self._path()):
            n = sum(1 for q in configs if q is f._config) if f._config is not None else 0
            if n:
                m = m.dot(np.linalg.matrix_power(-f.twist_hat(), n))
            m = m.dot(f.lg_inv())
        return m

    def g_dq(self, q1):
        return self._g_n((q1,)) if self._check((q1,)) else np.zeros((4, 4))

    def g_dqdq(self, q1, q2):
        return self._g_n((q1, q2)) if self._check((q1, q2)) else np.zeros((4, 4))

    def g_dqdqdq(self, q1, q2, q3):
        return self._g_n((q1, q2, q3)) if self._check((q1, q2, q3)) else np.zeros((4, 4))

    def g_dqdqdqdq(self, q1, q2, q3, q4):
        return self._g_n((q1, q2, q3, q4)) if self._check((q1, q2, q3, q4)) else np.zeros((4, 4))

    def g_inv(self):
        return self._g_inv_n(())

    def g_inv_dq(self, q1):
        return self._g_inv_n((q1,)) if self._check((q1,)) else np.zeros((4, 4))

    def g_inv_dqdq(self, q1, q2):
        return self._g_inv_n((q1, q2)) if self._check((q1, q2)) else np.zeros((4, 4))

    def p_dqdq(self, q1, q2):
        return self.g_dqdq(q1, q2)[:, 3].copy()

    def p_dqdqdq(self, q1, q2, q3):
        return self.g_dqdqdq(q1, q2, q3)[:, 3].copy()

    def p_dqdqdqdq(self, q1, q2, q3, q4):
        return self.g_dqdqdqdq(q1, q2, q3, q4)[:, 3].copy()

    # body velocity vb = g^-1 dg/dt = sum_k (g^-1 g_dq(k)) dq_k as a 4x4 se(3) matrix, and its derivatives by the
    # product rule over the subsets of the differentiation variables
    def _vb_term(self, k, configs):
        """d^n / d(configs) of g^-1 g_dq(k)."""
        n = len(configs)
        out = np.zeros((4, 4))
        for mask in range(1 << n):
            left = tuple(configs[i] for i in range(n) if mask >> i & 1)
            right = tuple(configs[i] for i in range(n) if not mask >> i & 1)
            out += self._g_inv_n(left).dot(self._g_n((k,) + right))
        return out

    def _driving(self):
        return [f._config for f in self._path() if f._config is not None]

    def _vb_n(self, configs):
        if not self._check(configs):
            return np.zeros((4, 4))
        out = np.zeros((4, 4))
        for k in self._driving():
            out += self._vb_term(k, tuple(configs)) * k.dq
        return out

    def _vb_ddq_n(self, dq1, configs):
        if not self._check((dq1,) + tuple(configs)):
            return np.zeros((4, 4))
        return self._vb_term(dq1, tuple(configs))

    def vb(self):
        return self._vb_n(())

    def vb_dq(self, q1):
        return self._vb_n((q1,))

    def vb_dqdq(self, q1, q2):
        return self._vb_n((q1, q2))

    def vb_dqdqdq(self, q1, q2, q3):
        return self._vb_n((q1, q2, q3))

    def vb_ddq(self, dq1):
        return self._vb_ddq_n(dq1, ())

    def vb_ddqdq(self, dq1, q2):
        return self._vb_ddq_n(dq1, (q2,))

    def vb_ddqdqdq(self, dq1, q2, q3):
        return self._vb_ddq_n(dq1, (q2, q3))

    def vb_ddqdqdqdq(self, dq1, q2, q3, q4):
        return self._vb_ddq_n(dq1, (q2, q3, q4))

    def g(self):
        chain = []
        f = self
        while f is not None:
            chain.append(f)
            f = f._parent
        m = np.eye(4)
        for f in reversed(chain):
            m = m.dot(f.lg())
        return m

    def p(self):
        return self.g()[:, 3].copy()

    def p_dq(self, config):
        """d p / d q (host, setup only): axis x (p - p_joint) for a rotary joint on the path from the world to
        this frame, the joint axis for a prismatic one, zero if the frame does not depend on `config`."""
        f = self
        while f is not None and f._config is not config:
            f = f._parent
        if f is None or config is None:
            return np.zeros(4)
        gj = f.g()
        kind = f._transform
        out = np.zeros(4)
        if kind in (TX, TY, TZ):
            out[:3] = gj[:3, (TX, TY, TZ).index(kind)]
        else:
            axis = gj[:3, (RX, RY, RZ).index(kind)]
            out[:3] = np.cross(axis, self.p()[:3] - gj[:3, 3])
        return out
