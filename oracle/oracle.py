"""ctypes harness around oracle/libtreporacle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under trep_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libtreporacle.so")
    src = os.path.join(_HERE, "trep_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libtreporacle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.to_create.restype = ctypes.c_void_p
        L.to_create.argtypes = [ctypes.c_void_p]
        L.to_destroy.argtypes = [ctypes.c_void_p]
        L.to_array.restype = ctypes.POINTER(ctypes.c_double)
        L.to_array.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
        L.to_set_times.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
        L.to_get_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.to_set_tolerance.argtypes = [ctypes.c_void_p, ctypes.c_double]
        L.to_solve_DEL.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.to_calc_p2.argtypes = [ctypes.c_void_p]
        L.to_calc_f.argtypes = [ctypes.c_void_p]
        L.to_calc_deriv1.argtypes = [ctypes.c_void_p]
        L.to_calc_deriv2.argtypes = [ctypes.c_void_p]
        L.to_step.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.to_rollout.restype = ctypes.c_int64
        L.to_rollout.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_int]
        L.to_dynamics.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 6
        L.to_energy.argtypes = [ctypes.c_void_p] * 4
        L.to_lagrangian.argtypes = [ctypes.c_void_p] * 5
        L.to_dynamics_deriv1.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 12
        _LIB = L
    return _LIB


class OracleError(Exception):
    pass


class OracleMVI(object):
    """Single-trajectory MidpointVI with the reference's semantics (midpointvi.py:138-201)."""

    def __init__(self, desc, tolerance=1e-10):
        self._L = lib()
        self._desc = desc
        self._h = self._L.to_create(ctypes.addressof(desc.struct))
        self.nq, self.nd, self.nk = int(desc.n_configs), int(desc.n_dyn), int(desc.n_kin)
        self.nu, self.nc = int(desc.n_inputs), int(desc.n_constraints)
        self._L.to_set_tolerance(self._h, tolerance)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.to_destroy(self._h)
            self._h = None

    def arr(self, name, shape=None):
        n = ctypes.c_int64()
        p = self._L.to_array(self._h, name.encode(), ctypes.byref(n))
        if n.value < 0:
            raise KeyError(name)
        if n.value == 0:
            a = np.zeros(0)
        else:
            a = np.ctypeslib.as_array(p, shape=(n.value,))
        return a.reshape(shape) if shape is not None else a

    def times(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        self._L.to_get_times(self._h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    t1 = property(lambda self: self.times()[0])
    t2 = property(lambda self: self.times()[1])

    def set_times(self, t1, t2):
        self._L.to_set_times(self._h, t1, t2)

    def _vec(name):
        def getter(self):
            return self.arr(name).copy()

        def setter(self, value):
            a = self.arr(name)
            if a.size:
                a[:] = np.asarray(value, dtype=float)
        return property(getter, setter)

    q1 = _vec("q1")
    q2 = _vec("q2")
    p1 = _vec("p1")
    p2 = _vec("p2")
    u1 = _vec("u1")
    lambda1 = _vec("lambda1")
    del _vec

    def initialize_from_state(self, t1, q1, p1, lambda1=None):
        self.set_times(t1, t1)
        self.q1 = q1
        self.p1 = p1
        self.q2 = q1
        self.p2 = p1
        self.lambda1 = np.zeros(self.nc) if lambda1 is None else lambda1

    def initialize_from_configs(self, t0, q0, t1, q1, lambda1=None):
        self.set_times(t0, t1)
        self.q1 = q0
        self.q2 = q1
        self._L.to_calc_p2(self._h)
        self.lambda1 = np.zeros(self.nc) if lambda1 is None else lambda1

    def step(self, t2, u1=(), k2=(), max_iterations=200, q2_hint=None, lambda1_hint=None):
        u1 = np.ascontiguousarray(u1, dtype=float)
        k2 = np.ascontiguousarray(k2, dtype=float)
        assert u1.shape == (self.nu,) and k2.shape == (self.nk,)
        if q2_hint is None and lambda1_hint is None:
            it = self._L.to_step(self._h, t2, u1.ctypes.data, k2.ctypes.data, max_iterations)
        else:
            self.q1 = self.q2
            self.p1 = self.p2
            self.u1 = u1
            q2 = self.arr("q2")
            q2[self.nd:] = k2
            self.set_times(self.t2, t2)
            if q2_hint is not None:
                q2[:self.nd] = np.asarray(q2_hint)[:self.nd]
            if lambda1_hint is not None:
                self.lambda1 = lambda1_hint
            it = self._L.to_solve_DEL(self._h, max_iterations)
        if it < 0:
            raise OracleError("not converged" if it == -1 else "singular")
        return it

    def solve_DEL(self, max_iterations=200):
        it = self._L.to_solve_DEL(self._h, max_iterations)
        if it < 0:
            raise OracleError("not converged" if it == -1 else "singular")
        return it

    def calc_f(self):
        self._L.to_calc_f(self._h)
        return self.arr("f").copy()

    def calc_p2(self):
        self._L.to_calc_p2(self._h)

    def calc_deriv1(self):
        if self._L.to_calc_deriv1(self._h):
            raise OracleError("singular")

    def calc_deriv2(self):
        if self._L.to_calc_deriv2(self._h):
            raise OracleError("singular")

    def deriv1(self, name):
        """First-derivative array in the reference's C layout [derivative variable][output]."""
        out = {"q2": self.nd, "p2": self.nd, "l1": self.nc}[name[:2]]
        rows = {"dq1": self.nq, "dp1": self.nd, "du1": self.nu, "dk2": self.nk}[name[3:]]
        return self.arr(name, (rows, out)).copy()

    def deriv2(self, name):
        out = {"q2": self.nd, "p2": self.nd, "l1": self.nc}[name[:2]]
        cnt = {"dq1": self.nq, "dp1": self.nd, "du1": self.nu, "dk2": self.nk}
        a, b = name[3:6], name[6:9]
        return self.arr(name, (cnt[a], cnt[b], out)).copy()

    def dynamics(self, q, dq, u=None, ddqk=None):
        """Continuous dynamics (system.c:749-893): returns (ddq [nd], lambda [nc])."""
        q = np.ascontiguousarray(q, dtype=float)
        dq = np.ascontiguousarray(dq, dtype=float)
        u = np.zeros(max(self.nu, 1)) if u is None else np.ascontiguousarray(np.append(u, 0.0), dtype=float)
        ddqk = np.zeros(max(self.nk, 1)) if ddqk is None else np.ascontiguousarray(np.append(ddqk, 0.0), dtype=float)
        f, lam = np.zeros(max(self.nd, 1)), np.zeros(max(self.nc, 1))
        if self._L.to_dynamics(self._h, q.ctypes.data, dq.ctypes.data, u.ctypes.data, ddqk.ctypes.data,
                               f.ctypes.data, lam.ctypes.data):
            raise OracleError("singular")
        return f[:self.nd], lam[:self.nc]

    def energy(self, q, dq):
        """(kinetic, potential) energy at (q, dq) (system.c:78-127)."""
        q = np.ascontiguousarray(q, dtype=float)
        dq = np.ascontiguousarray(dq, dtype=float)
        out = np.zeros(2)
        self._L.to_energy(self._h, q.ctypes.data, dq.ctypes.data, out.ctypes.data)
        return float(out[0]), float(out[1])

    def lagrangian(self, q, dq):
        """(L_dq [nq], L_ddq [nq], L_dqdq, L_ddqdq (dq row, q column), L_ddqddq [nq][nq]) at (q, dq)."""
        q = np.ascontiguousarray(q, dtype=float)
        dq = np.ascontiguousarray(dq, dtype=float)
        o1, o2 = np.zeros((2, self.nq)), np.zeros((3, self.nq, self.nq))
        self._L.to_lagrangian(self._h, q.ctypes.data, dq.ctypes.data, o1.ctypes.data, o2.ctypes.data)
        return o1[0], o1[1], o2[0], o2[1], o2[2]

    def dynamics_deriv1(self, q, dq, u=None, ddqk=None):
        """First derivatives of the continuous dynamics (system.c:912-1299) in the layout of the reference's accessors
        System.f_dq() ... lambda_du(): dict of [output][derivative variable] arrays."""
        q = np.ascontiguousarray(q, dtype=float)
        dq = np.ascontiguousarray(dq, dtype=float)
        u = np.zeros(max(self.nu, 1)) if u is None else np.ascontiguousarray(np.append(u, 0.0), dtype=float)
        ddqk = np.zeros(max(self.nk, 1)) if ddqk is None else np.ascontiguousarray(np.append(ddqk, 0.0), dtype=float)
        rows = {"dq": self.nq, "ddq": self.nq, "dk": self.nk, "du": self.nu}
        arrs = {}
        for pre, width in (("f", self.nd), ("l", self.nc)):
            for var in ("dq", "ddq", "dk", "du"):
                arrs[pre + "_" + var] = np.zeros((rows[var], width)) if rows[var] * width else np.zeros((max(rows[var], 1), max(width, 1)))
        order = ["f_dq", "f_ddq", "f_dk", "f_du", "l_dq", "l_ddq", "l_dk", "l_du"]
        if self._L.to_dynamics_deriv1(self._h, q.ctypes.data, dq.ctypes.data, u.ctypes.data, ddqk.ctypes.data,
                                      *[arrs[n].ctypes.data for n in order]):
            raise OracleError("singular")
        out = {}
        for pre, width, name in (("f", self.nd, "f"), ("l", self.nc, "lam")):
            for var, key in (("dq", "dq"), ("ddq", "ddq"), ("dk", "dddk"), ("du", "du")):
                out["%s_%s" % (name, key)] = arrs[pre + "_" + var][:rows[var], :width].T.copy()
        return out

    def rollout(self, n_steps, dt, U=None, K=None, want_X=True, max_iterations=200):
        nX = self.nq + self.nd + self.nk
        U = None if U is None or self.nu == 0 else np.ascontiguousarray(U, dtype=float)
        K = None if K is None or self.nk == 0 else np.ascontiguousarray(K, dtype=float)
        X = np.zeros((n_steps + 1, nX)) if want_X else None
        tot = self._L.to_rollout(self._h, n_steps, dt,
                                 None if U is None else U.ctypes.data,
                                 None if K is None else K.ctypes.data,
                                 None if X is None else X.ctypes.data, max_iterations)
        if tot < 0:
            raise OracleError("rollout failed (%d)" % tot)
        return X, int(tot)
