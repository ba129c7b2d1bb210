"""Generalized coordinates and force inputs (host-side model objects).

Mirrors the public surface of the reference's ``trep.Config``
(/root/reference/trep/config.py:4-122) and ``trep.Input``
(/root/reference/trep/finput.py:4-55).  These are plain Python objects: the
numbers that matter to the device path (index, k_index, config_gen, masses) are
assigned by ``System._sync`` and flattened by ``trep_amd.descriptor``.
"""


class Config(object):
    """One generalized coordinate; dynamic by default, kinematic on request."""

    def __init__(self, system, name=None, kinematic=False):
        self._system = system
        self.name = name
        self._kinematic = bool(kinematic)
        self._q = 0.0
        self._dq = 0.0
        self._ddq = 0.0
        self._index = -1
        self._k_index = -1
        self._config_gen = -1
        self._masses = tuple()
        system._register_config(self)

    def __repr__(self):
        return "<Config %r %f %f %f>" % (self.name or id(self), self.q, self.dq, self.ddq)

    system = property(lambda self: self._system)
    kinematic = property(lambda self: self._kinematic)

    @property
    def frame(self):
        for f in self.system.frames:
            if f.config is self:
                return f
        return None

    @property
    def q(self):
        return self._q

    @q.setter
    def q(self, value):
        self._q = float(value)

    @property
    def dq(self):
        return self._dq

    @dq.setter
    def dq(self, value):
        self._dq = float(value)

    @property
    def ddq(self):
        return self._ddq

    @ddq.setter
    def ddq(self, value):
        self._ddq = float(value)

    @property
    def index(self):
        self._system._sync()
        return self._index

    @property
    def k_index(self):
        self._system._sync()
        return self._k_index

    @property
    def config_gen(self):
        self._system._sync()
        return self._config_gen

    @property
    def masses(self):
        self._system._sync()
        return self._masses


class Input(object):
    """A scalar force input, created by a Force (finput.py:17-55)."""

    def __init__(self, system, name=None):
        self._system = system
        self.name = name
        self._u = 0.0
        self._index = -1
        self._force = None
        system._register_input(self)

    def __repr__(self):
        return "<Input %r %f>" % (self.name or id(self), self.u)

    system = property(lambda self: self._system)
    force = property(lambda self: self._force)

    @property
    def u(self):
        return self._u

    @u.setter
    def u(self, v):
        self._u = float(v)

    @property
    def index(self):
        self._system._sync()
        return self._index
