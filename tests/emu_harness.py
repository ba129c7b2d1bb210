"""ctypes driver for tests/emu/libtrepamd_emu.so: the device kernel source compiled for the host
(TEAM = 1).  Test infrastructure only; lets the CPU suite exercise the kernel logic without a GPU."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = None

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)


class RunArgs(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int), ("n_steps", ctypes.c_int), ("max_iterations", ctypes.c_int),
                ("mode", ctypes.c_int), ("predictor", ctypes.c_int),
                ("dt", ctypes.c_double), ("t1", ctypes.c_double), ("t2", ctypes.c_double),
                ("tolerance", ctypes.c_double),
                ("q1", _D), ("q2", _D), ("p1", _D), ("p2", _D), ("lam", _D), ("u1", _D),
                ("U", _D), ("K", _D), ("q2_hint", _D), ("lam_hint", _D), ("Kproj", _D), ("bX", _D), ("bU", _D), ("Uout", _D), ("group_size", ctypes.c_int), ("group_map", _I), ("X", _D), ("f_out", _D),
                ("d1", _D * 12), ("A_out", _D), ("B_out", _D), ("z", _D), ("hz", _D), ("iters", _I), ("status", _I), ("prof_out", ctypes.c_void_p), ("dt_steps", _D), ("dt_period", ctypes.c_int), ("exact_pivot", ctypes.c_int),
                ("zl", _D), ("dq_in", _D), ("ddqk_in", _D), ("ddq_out", _D), ("lam_out", _D), ("g1", _D * 8), ("energy_out", _D), ("lag1_out", _D), ("lag2_out", _D), ("mirror", _D),
                ("remap_len", ctypes.c_int), ("remap_stride", ctypes.c_int), ("remap_off", ctypes.c_int), ("remap_count", ctypes.c_int),
                ("fallbacks", _I), ("seed1", _I), ("seed2", _I)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "emu", "libtrepamd_emu.so")
        srcs = [os.path.join(_HERE, "emu", "emu.cpp"),
                os.path.join(_ROOT, "trep_amd", "csrc", "mvi_core.hpp"),
                os.path.join(_ROOT, "trep_amd", "csrc", "program.hpp"),
                os.path.join(_ROOT, "trep_amd", "csrc", "bbd.hpp"),
                os.path.join(_ROOT, "trep_amd", "csrc", "dual.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, srcs[0]], check=True)
        L = ctypes.CDLL(so)
        L.emu_create.restype = ctypes.c_void_p
        L.emu_create.argtypes = [ctypes.c_void_p]
        L.emu_destroy.argtypes = [ctypes.c_void_p]
        L.emu_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.emu_run_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.emu_lds_doubles.argtypes = [ctypes.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(_D)


class EmuBatch(object):
    """Batch state + the three kernel modes, mirroring what tg_batch_* does on the device."""

    def __init__(self, desc, batch, tolerance=1e-10):
        self.L = lib()
        self.h = self.L.emu_create(ctypes.addressof(desc.struct))
        assert self.h
        self.desc = desc
        self.B = batch
        self.nq, self.nd, self.nk = int(desc.n_configs), int(desc.n_dyn), int(desc.n_kin)
        self.nu, self.nc = int(desc.n_inputs), int(desc.n_constraints)
        self.nX = self.nq + self.nd + self.nk
        z = lambda w: np.zeros((batch, w))
        self.q1, self.q2, self.p1, self.p2, self.lam, self.u1 = z(self.nq), z(self.nq), z(self.nd), z(self.nd), z(self.nc), z(self.nu)
        self.iters = np.zeros(batch, dtype=np.int32)
        self.status = np.zeros(batch, dtype=np.int32)
        self.t1 = self.t2 = 0.0
        self.tol = tolerance

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emu_destroy(self.h)
            self.h = None

    def _args(self, mode, n_steps=0, dt=0.0, U=None, K=None, X=None, f=None, q2_hint=None, lam_hint=None, max_it=200):
        a = RunArgs()
        a.batch, a.n_steps, a.max_iterations, a.mode = self.B, n_steps, max_it, mode
        a.group_size = 1
        a.dt, a.t1, a.t2, a.tolerance = dt, self.t1, self.t2, self.tol
        a.q1, a.q2, a.p1, a.p2, a.lam, a.u1 = _p(self.q1), _p(self.q2), _p(self.p1), _p(self.p2), _p(self.lam), _p(self.u1)
        a.U, a.K, a.X, a.f_out, a.q2_hint, a.lam_hint = _p(U), _p(K), _p(X), _p(f), _p(q2_hint), _p(lam_hint)
        a.iters = self.iters.ctypes.data_as(_I)
        a.status = self.status.ctypes.data_as(_I)
        return a

    def _run(self, a, seeds=None):
        """The kernel body over the batch; seeds = (seed1,) or (seed1, seed2) [B] int32 arrays: the forward-mode kernel (run_forward) instead,
        whose outputs are the derivatives along those input variables (numbered q | dq | ddq_k | u)."""
        if seeds is None:
            self.L.emu_run(self.h, ctypes.byref(a))
            return
        keep = [np.ascontiguousarray(s_, dtype=np.int32) for s_ in seeds]
        assert all(k.shape == (self.B,) for k in keep)
        a.seed1 = keep[0].ctypes.data_as(_I)
        if len(keep) > 1:
            a.seed2 = keep[1].ctypes.data_as(_I)
        self.L.emu_run_forward(self.h, ctypes.byref(a), len(keep))

    def initialize_from_configs(self, t0, Q0, t1, Q1):
        self.t1, self.t2 = t0, t1
        self.q1[:], self.q2[:] = Q0, Q1
        a = self._args(1)
        self.L.emu_run(self.h, ctypes.byref(a))
        self.lam[:] = 0.0

    def calc_f(self):
        f = np.zeros((self.B, self.nd + self.nc))
        a = self._args(2, f=f)
        self.L.emu_run(self.h, ctypes.byref(a))
        return f

    def deriv1(self):
        """Returns {name: [B][var][out]} for the 12 first-derivative arrays of the last solved step."""
        rows = {"dq1": self.nq, "dp1": self.nd, "du1": self.nu, "dk2": self.nk}
        outs = {}
        a = self._args(3)
        for oi, (pre, width) in enumerate((("q2", self.nd), ("p2", self.nd), ("l1", self.nc))):
            for ki, var in enumerate(("dq1", "dp1", "du1", "dk2")):
                arr = np.zeros((self.B, rows[var], width))
                outs["%s_%s" % (pre, var)] = arr
                a.d1[4 * oi + ki] = arr.ctypes.data_as(_D) if arr.size else ctypes.cast(0, _D)
        self.L.emu_run(self.h, ctypes.byref(a))
        return outs

    def linearize(self):
        """DSystem.fdx / fdu of the last solved step, written by the deriv1 kernel in A/B form."""
        nU = self.nu + self.nk
        A = np.full((self.B, self.nX, self.nX), np.nan)
        Bm = np.full((self.B, self.nX, nU), np.nan)
        a = self._args(3)
        a.A_out, a.B_out = _p(A), _p(Bm)
        self.L.emu_run(self.h, ctypes.byref(a))
        return A, Bm

    def rollout_closed_loop(self, n_steps, dt, Kproj, bX, bU, group_size=1):
        """U_k = bU_k - Kproj_k (X_k - bX_k) in-kernel; returns (X [B][N+1][nX], U [B][N][nU])."""
        Kproj = np.ascontiguousarray(Kproj, dtype=float)
        bX = np.ascontiguousarray(bX, dtype=float)
        bU = np.ascontiguousarray(bU, dtype=float)
        X = np.zeros((self.B, n_steps + 1, self.nX))
        Uo = np.zeros((self.B, n_steps, self.nu + self.nk))
        a = self._args(0, n_steps, dt, None, None, X)
        a.Kproj, a.bX, a.bU, a.Uout, a.group_size = _p(Kproj), _p(bX), _p(bU), _p(Uo), group_size
        self.L.emu_run(self.h, ctypes.byref(a))
        self.t1, self.t2 = self.t2 + (n_steps - 1) * dt, self.t2 + n_steps * dt
        return X, Uo

    def dynamics(self, Q, dQ, U=None, ddK=None):
        """Continuous dynamics of every trajectory: returns (ddq [B][nd], lambda [B][nc], status [B])."""
        Q = np.ascontiguousarray(Q, dtype=float)
        dQ = np.ascontiguousarray(dQ, dtype=float)
        U = np.zeros((self.B, self.nu)) if U is None else np.ascontiguousarray(U, dtype=float)
        ddK = np.zeros((self.B, self.nk)) if ddK is None else np.ascontiguousarray(ddK, dtype=float)
        ddq, lam = np.zeros((self.B, self.nd)), np.zeros((self.B, self.nc))
        a = self._args(5)
        a.q1 = a.q2 = _p(Q)
        a.u1 = _p(U)
        a.dq_in, a.ddqk_in, a.ddq_out, a.lam_out = _p(dQ), _p(ddK), _p(ddq), _p(lam)
        self.L.emu_run(self.h, ctypes.byref(a))
        return ddq, lam, self.status.copy()

    def energy(self, Q, dQ):
        """[B][2]: kinetic and potential energy of every state."""
        Q = np.ascontiguousarray(Q, dtype=float)
        dQ = np.ascontiguousarray(dQ, dtype=float)
        out = np.zeros((self.B, 2))
        a = self._args(7)
        a.q1 = a.q2 = _p(Q)
        a.dq_in, a.energy_out = _p(dQ), _p(out)
        self.L.emu_run(self.h, ctypes.byref(a))
        return out

    def lagrangian(self, Q, dQ, seeds=None):
        """(L1 [B][2][nq], L2 [B][3][nq][nq]): first and second derivatives of the Lagrangian of every state (seeds: their derivatives)."""
        Q = np.ascontiguousarray(Q, dtype=float)
        dQ = np.ascontiguousarray(dQ, dtype=float)
        o1, o2 = np.zeros((self.B, 2, self.nq)), np.zeros((self.B, 3, self.nq, self.nq))
        a = self._args(8)
        a.q1 = a.q2 = _p(Q)
        a.dq_in, a.lag1_out, a.lag2_out = _p(dQ), _p(o1), _p(o2)
        self._run(a, seeds)
        return o1, o2

    def dynamics_deriv1(self, Q, dQ, U=None, ddK=None, seeds=None):
        """First derivatives of the continuous dynamics, in the layout of the reference's accessors
        (System.f_dq() ...): dict of [B][output][derivative variable] arrays."""
        Q = np.ascontiguousarray(Q, dtype=float)
        dQ = np.ascontiguousarray(dQ, dtype=float)
        U = np.zeros((self.B, self.nu)) if U is None else np.ascontiguousarray(U, dtype=float)
        ddK = np.zeros((self.B, self.nk)) if ddK is None else np.ascontiguousarray(ddK, dtype=float)
        rows = [self.nq, self.nq, self.nk, self.nu]
        names = ["dq", "ddq", "dddk", "du"]
        a = self._args(6)
        a.q1 = a.q2 = _p(Q)
        a.u1 = _p(U)
        a.dq_in, a.ddqk_in = _p(dQ), _p(ddK)
        arrs = {}
        for g in range(8):
            width = self.nd if g < 4 else self.nc
            arr = np.zeros((self.B, rows[g & 3], width))
            arrs[("f_" if g < 4 else "lam_") + names[g & 3]] = arr
            a.g1[g] = arr.ctypes.data_as(_D) if arr.size else ctypes.cast(0, _D)
        self._run(a, seeds)
        return dict((k, np.swapaxes(v, 1, 2)) for k, v in arrs.items()), self.status.copy()

    def deriv2z(self, Z, ZL=None):
        """HZ [B][R][R]: second derivatives of the step map contracted with z = Z[b] (nX) and, optionally, of
        lambda1 contracted with ZL[b] (nc)."""
        R = self.nq + self.nd + self.nu + self.nk
        Z = np.ascontiguousarray(Z, dtype=float)
        HZ = np.zeros((self.B, R, R))
        a = self._args(4)
        a.z, a.hz = _p(Z), _p(HZ)
        if ZL is not None:
            ZL = np.ascontiguousarray(ZL, dtype=float)
            a.zl = _p(ZL)
        self.L.emu_run(self.h, ctypes.byref(a))
        return HZ

    def rollout(self, n_steps, dt, U=None, K=None, want_X=True, q2_hint=None, lam_hint=None, dts=None):
        """dts: optional step sizes, one per step (non-uniform time base: RunArgs.dt_steps)."""
        U = None if U is None else np.ascontiguousarray(U, dtype=float)
        K = None if K is None else np.ascontiguousarray(K, dtype=float)
        X = np.zeros((self.B, n_steps + 1, self.nX)) if want_X else None
        a = self._args(0, n_steps, dt, U, K, X, q2_hint=q2_hint, lam_hint=lam_hint)
        if dts is not None:
            dts = np.ascontiguousarray(dts, dtype=float)
            a.dt_steps, a.dt_period = _p(dts), 0
        self.L.emu_run(self.h, ctypes.byref(a))
        if dts is None:
            self.t1, self.t2 = self.t2 + (n_steps - 1) * dt, self.t2 + n_steps * dt
        else:
            self.t1, self.t2 = self.t2 + float(dts[:n_steps - 1].sum()), self.t2 + float(dts[:n_steps].sum())
        return X
