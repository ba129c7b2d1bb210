"""MidpointVI: the midpoint variational integrator, executed on the MI355X.

``BatchMidpointVI`` advances B independent trajectories of one ``System`` at
once on a HIP device; ``MidpointVI`` is the B = 1 shell that keeps the reference
API (/root/reference/trep/midpointvi.py:19-332: initialize_from_state /
initialize_from_configs / step / calc_f, the t/q/p/u/lambda properties, v2) so
the reference's example scripts run unmodified against this package.

All numerical work happens in libtrepamd.so (HIP, gfx950) through the C ABI of
include/trep_amd.h.  State lives in device memory; the properties copy it
in/out.  There is no CPU execution path: without the library or without a GPU
every call raises.
"""
import os

import numpy as np

from . import _lib
from .descriptor import flatten
from .errors import ConvergenceError


class BatchMidpointVI(object):
    """B trajectories of ``system`` on HIP device ``device`` (state arrays are [B][width])."""

    def __init__(self, system, batch, tolerance=1e-10, device=0, specialize="auto"):
        """specialize: "auto" (default) uses a system-specialised rollout kernel if one has been built for this system
        (trep_amd/specialize.py; never compiles implicitly), True builds one if needed, False keeps the generic kernel."""
        self._system = system
        self._specialize_mode = specialize
        self._specialized = None
        self._batch = int(batch)
        self._device = int(device)
        self._L = _lib.lib()
        _lib.require_device()
        self._desc = flatten(system)
        self._structure_version = system._structure_version
        self._sys_h = self._L.tg_system_create(self._desc.byref())
        if not self._sys_h:
            raise _lib.LibraryError(self._L.tg_last_error().decode())
        self._h = self._L.tg_batch_create(self._sys_h, self._batch, self._device)
        if not self._h:
            msg = self._L.tg_last_error().decode()
            self._L.tg_system_destroy(self._sys_h)
            self._sys_h = None
            raise _lib.LibraryError(msg)
        self.tolerance = tolerance
        self._owned_dev = []
        if specialize is True:
            self.specialize(build=True)
        elif specialize == "auto" and os.environ.get("TREPAMD_NO_SPECIALIZE") is None:
            self.specialize(build=False)

    def specialize(self, build=True):
        """Use a rollout kernel compiled for exactly this system (trep_amd/specialize.py): same results, fewer
        instructions.  With build=False only an already cached specialisation is used (returns False if there is
        none).  A later structure / parameter change of the system drops back to the generic kernel."""
        from . import specialize as _spec
        if not build and not _spec.is_built(self._system):
            return False
        path = os.environ.get("TREPAMD_SPEC_OVERRIDE") or _spec.build(self._system)   # override: A/B experiments only
        _lib.check(self._L.tg_batch_load_specialized(self._h, path.encode()))
        self._specialized = path
        return True

    def refresh(self):
        """Re-flatten the system if it changed since the device program was built.  The reference treats parameter
        writes (Gravity.gravity, spring constants, damping, Distance.distance ...) as plain attribute updates and its
        integrator keeps its state across them (trep/potentials/gravity.py, trep/midpointvi.py:25-136 only
        re-allocates on *structure* changes); here every change re-builds the few-kB device schedule, and the batch
        state (times, q1, q2, p1, p2, u1, lambda1, tolerance, predictor) is carried over whenever the sizes are
        unchanged.  Returns True if a rebuild happened."""
        if self._structure_version == self._system._structure_version:
            return False
        old_sizes = (self.nq, self.nd, self.nk, self.nu, self.nc)
        desc = flatten(self._system)
        sys_h = self._L.tg_system_create(desc.byref())
        if not sys_h:
            raise _lib.LibraryError(self._L.tg_last_error().decode())
        h = self._L.tg_batch_create(sys_h, self._batch, self._device)
        if not h:
            msg = self._L.tg_last_error().decode()
            self._L.tg_system_destroy(sys_h)
            raise _lib.LibraryError(msg)
        keep = None
        if old_sizes == (int(desc.n_configs), int(desc.n_dyn), int(desc.n_kin), int(desc.n_inputs), int(desc.n_constraints)):
            keep = (self.times(), [getattr(self, n) for n in ("q1", "q2", "p1", "p2", "u1", "lambda1")], self.tolerance,
                    self.predictor, self.exact_pivot)
        self._L.tg_batch_destroy(self._h)
        self._L.tg_system_destroy(self._sys_h)
        self._desc, self._sys_h, self._h = desc, sys_h, h
        self._structure_version = self._system._structure_version
        self._specialized = None
        if self._specialize_mode is True or (self._specialize_mode == "auto" and os.environ.get("TREPAMD_NO_SPECIALIZE") is None):
            self.specialize(build=False)      # a cached specialisation of the NEW schedule, if there is one
        if keep is not None:
            (t1, t2), fields, tol, pred, exact = keep
            self.set_times(t1, t2)
            for n, v in zip(("q1", "q2", "p1", "p2", "u1", "lambda1"), fields):
                setattr(self, n, v)
            self.tolerance = tol
            self.predictor = pred
            self.exact_pivot = exact
        ss = getattr(self, "_step_sizes", None)
        if ss is not None:
            self.set_step_sizes(ss[0], ss[1])
        return True

    def close(self):
        if getattr(self, "_h", None):
            for p in self._owned_dev:
                self._L.tg_device_free(self._device, p)
            self._owned_dev = []
            self._L.tg_batch_destroy(self._h)
            self._h = None
        if getattr(self, "_sys_h", None):
            self._L.tg_system_destroy(self._sys_h)
            self._sys_h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_step_sizes(self, dts=None, by_trajectory=False):
        """Non-uniform time base (include/trep_amd.h, tg_batch_set_step_sizes): `dts` replaces the scalar dt of the rollouts
        (step k uses dts[k]) or, with by_trajectory, of a one-step batch (trajectory t uses dts[t % len(dts)]).  None clears."""
        if dts is None:
            _lib.check(self._L.tg_batch_set_step_sizes(self._h, 0, None, 0))
            self._step_sizes = None
            return
        d = np.ascontiguousarray(dts, dtype=np.float64).reshape(-1)
        _lib.check(self._L.tg_batch_set_step_sizes(self._h, len(d), d.ctypes.data, 1 if by_trajectory else 0))
        self._step_sizes = (d.copy(), bool(by_trajectory))

    @staticmethod
    def _dt_argument(dt, n_steps):
        """(scalar dt for the C call, step-size list or None): rollouts accept a scalar or one step size per step."""
        if np.ndim(dt) == 0:
            return float(dt), None
        d = np.ascontiguousarray(dt, dtype=np.float64).reshape(-1)
        if len(d) != n_steps:
            raise ValueError("expected %d step sizes, got %d" % (n_steps, len(d)))
        if np.allclose(d, d[0], rtol=1e-13, atol=0.0):
            return float(d[0]), None
        return float(d[0]), d

    @property
    def exact_pivot(self):
        """False (default): pivot candidates of the Newton solve are ranked in single precision; True: the reference's
        pivot rule bit for bit (include/trep_amd.h, tg_batch_set_pivot_rule), about 9 % slower."""
        return getattr(self, "_exact_pivot", False)

    @exact_pivot.setter
    def exact_pivot(self, value):
        _lib.check(self._L.tg_batch_set_pivot_rule(self._h, 1 if value else 0))
        self._exact_pivot = bool(value)

    # -- sizes ---------------------------------------------------------------------------
    system = property(lambda self: self._system)
    batch = property(lambda self: self._batch)
    device = property(lambda self: self._device)
    nq = property(lambda self: int(self._desc.n_configs))
    nd = property(lambda self: int(self._desc.n_dyn))
    nk = property(lambda self: int(self._desc.n_kin))
    nu = property(lambda self: int(self._desc.n_inputs))
    nc = property(lambda self: int(self._desc.n_constraints))
    nX = property(lambda self: self.nq + self.nd + self.nk)
    nU = property(lambda self: self.nu + self.nk)

    def info(self):
        out = (np.zeros(8, dtype=np.int32))
        _lib.check(self._L.tg_system_info(self._sys_h, out.ctypes.data_as(_lib._c_ip)))
        keys = ["team", "lds_bytes_per_trajectory", "joints", "levels", "bodies", "items", "pairs", "dh_items"]
        return dict(zip(keys, (int(v) for v in out)))

    MODES = {"rollout": 0, "calc_p2": 1, "calc_f": 2, "deriv1": 3, "deriv2z": 4}

    def kernel_info(self):
        """Which kernels this batch has launched (tg_batch_info): `spec_modes` = set of mode names with a specialised kernel
        loaded, `spec_launched` / `generic_launched` = mode names that have actually gone through a specialised / generic
        kernel, and the launch counts; `helper_waves` = wavefronts per trajectory in the loaded library's derivative kernels;
        `spec_library` the loaded file."""
        out = np.zeros(8, dtype=np.int32)
        _lib.check(self._L.tg_batch_info(self._h, out.ctypes.data_as(_lib._c_ip)))
        names = lambda bits: sorted(n for n, m in self.MODES.items() if (int(bits) >> m) & 1)
        return {"spec_modes": names(out[0]), "spec_launched": names(out[1]), "generic_launched": names(out[2]),
                "spec_launch_mask": int(out[1]), "generic_launch_mask": int(out[2]),
                "spec_launches": int(out[3]), "generic_launches": int(out[4]), "exact_pivot": bool(out[5]), "team": int(out[6]),
                "helper_waves": int(out[7]), "spec_library": self._specialized}

    @property
    def stream(self):
        """The batch's HIP stream (hipStream_t as an integer), e.g. for ``Communicator.wait_stream``."""
        return self._L.tg_batch_stream(self._h)

    # -- state ---------------------------------------------------------------------------
    @property
    def tolerance(self):
        return self._tolerance

    @tolerance.setter
    def tolerance(self, value):
        self._tolerance = float(value)
        _lib.check(self._L.tg_batch_set_tolerance(self._h, self._tolerance))

    def times(self):
        import ctypes
        a, b = ctypes.c_double(), ctypes.c_double()
        _lib.check(self._L.tg_batch_get_times(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def set_times(self, t1, t2):
        _lib.check(self._L.tg_batch_set_times(self._h, float(t1), float(t2)))

    def _get(self, field, width):
        out = np.zeros((self._batch, width))
        if width:
            _lib.check(self._L.tg_batch_get(self._h, field, out.ctypes.data))
        return out

    def _set(self, field, width, value):
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64), (self._batch, width)))
        if width:
            _lib.check(self._L.tg_batch_set(self._h, field, arr.ctypes.data))

    def _field(field, width_attr):
        def getter(self):
            return self._get(field, getattr(self, width_attr))

        def setter(self, value):
            self._set(field, getattr(self, width_attr), value)
        return property(getter, setter)

    q1 = _field(_lib.F_Q1, "nq")
    q2 = _field(_lib.F_Q2, "nq")
    p1 = _field(_lib.F_P1, "nd")
    p2 = _field(_lib.F_P2, "nd")
    u1 = _field(_lib.F_U1, "nu")
    lambda1 = _field(_lib.F_LAMBDA1, "nc")
    del _field

    # -- reference semantics, batched -------------------------------------------------------------
    def initialize_from_state(self, t1, q1, p1, lambda1=None):
        """midpointvi.py:138-153: (t2,q2,p2) <- (t1,q1,p1), lambda1 <- given or 0."""
        self.refresh()
        self.set_times(t1, t1)
        self.q1 = q1
        self.p1 = p1
        self.q2 = q1
        self.p2 = p1
        self.lambda1 = np.zeros((self._batch, self.nc)) if lambda1 is None else lambda1

    def initialize_from_configs(self, t0, q0, t1, q1, lambda1=None):
        """midpointvi.py:155-172: p2 = D2L2(q0, q1) computed on the device."""
        self.refresh()
        self.set_times(t0, t1)
        self.q1 = q0
        self.q2 = q1
        _lib.check(self._L.tg_batch_calc_p2(self._h))
        self.lambda1 = np.zeros((self._batch, self.nc)) if lambda1 is None else lambda1

    def calc_p2(self):
        self.refresh()
        _lib.check(self._L.tg_batch_calc_p2(self._h))

    def calc_f(self):
        self.refresh()
        out = np.zeros((self._batch, self.nd + self.nc))
        _lib.check(self._L.tg_batch_calc_f(self._h, out.ctypes.data))
        return out

    # -- first derivatives (reference MidpointVI_calc_deriv1, midpointvi.c:1100-1120) -----------------
    D1_NAMES = ["q2_dq1", "q2_dp1", "q2_du1", "q2_dk2", "p2_dq1", "p2_dp1", "p2_du1", "p2_dk2",
                "l1_dq1", "l1_dp1", "l1_du1", "l1_dk2"]

    def calc_deriv1(self):
        """Compute all twelve first-derivative arrays of the last solved step on the device."""
        self.refresh()
        _lib.check(self._L.tg_batch_deriv1(self._h))

    def deriv1(self, name):
        """[B][derivative variable][output] (the reference's C layout, trep.h:425-437)."""
        k = self.D1_NAMES.index(name)
        rows = (self.nq, self.nd, self.nu, self.nk)[k % 4]
        width = self.nc if k // 4 == 2 else self.nd
        out = np.zeros((self._batch, rows, width))
        if out.size:
            _lib.check(self._L.tg_batch_get(self._h, _lib.F_D1_BASE + k, out.ctypes.data))
        return out

    def deriv2_contract(self, Z, ZL=None):
        """Second derivatives contracted with Z [B][nX] over the output index: HZ [B][R][R] with the
        derivative variables ordered (q1[nq], p1[nd], u1[nu], k2[nk]).  HZ[b][A][B] =
        sum_o Z[b][o] q2_dAdB[A][B][o] + Z[b][nq+o] p2_dAdB[A][B][o] (what DSystem.fdxdx/fdxdu/fdudu use)."""
        self.refresh()
        R = self.nq + self.nd + self.nu + self.nk
        Z = _lib.as_f64(np.broadcast_to(np.asarray(Z, dtype=float), (self._batch, self.nX)), (self._batch, self.nX))
        HZ = np.zeros((self._batch, R, R))
        if ZL is None:
            _lib.check(self._L.tg_batch_deriv2_contract(self._h, Z.ctypes.data, HZ.ctypes.data))
        else:   # additionally sum_c ZL[b][c] * lambda1_dAdB[A][B][c]
            ZL = _lib.as_f64(np.broadcast_to(np.asarray(ZL, dtype=float), (self._batch, self.nc)), (self._batch, self.nc))
            _lib.check(self._L.tg_batch_deriv2_contract_lambda(self._h, Z.ctypes.data, _lib.ptr(ZL), HZ.ctypes.data))
        return HZ

    def step(self, t2, u1=None, k2=None, max_iterations=200, q2_hint=None, lambda1_hint=None):
        """One MidpointVI.step for every trajectory.  Returns (iterations[B], status[B])."""
        self.refresh()
        B = self._batch
        u = None if self.nu == 0 else _lib.as_f64(np.broadcast_to(np.asarray(u1, dtype=float), (B, self.nu)), (B, self.nu))
        k = None if self.nk == 0 else _lib.as_f64(np.broadcast_to(np.asarray(k2, dtype=float), (B, self.nk)), (B, self.nk))
        qh = None if q2_hint is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(q2_hint, dtype=float)[..., :self.nd], (B, self.nd)))
        lh = None if lambda1_hint is None or self.nc == 0 else _lib.as_f64(
            np.broadcast_to(np.asarray(lambda1_hint, dtype=float), (B, self.nc)), (B, self.nc))
        iters = np.zeros(B, dtype=np.int32)
        status = np.zeros(B, dtype=np.int32)
        _lib.check(self._L.tg_batch_step(self._h, float(t2), _lib.ptr(u), _lib.ptr(k), _lib.ptr(qh), _lib.ptr(lh),
                                         int(max_iterations), iters.ctypes.data, status.ctypes.data))
        return iters, status

    # -- device-resident rollouts ------------------------------------------------------------------
    def device_array(self, host):
        """Upload a host array; returns a device pointer owned by this batch."""
        host = np.ascontiguousarray(host, dtype=np.float64)
        p = self._L.tg_device_alloc(self._device, max(host.nbytes, 8))
        if not p:
            raise _lib.LibraryError(self._L.tg_last_error().decode())
        if host.nbytes:
            _lib.check(self._L.tg_memcpy_h2d(self._device, p, host.ctypes.data, host.nbytes))
        self._owned_dev.append(p)
        return p

    def device_empty(self, n_doubles):
        p = self._L.tg_device_alloc(self._device, max(8 * int(n_doubles), 8))
        if not p:
            raise _lib.LibraryError(self._L.tg_last_error().decode())
        self._owned_dev.append(p)
        return p

    def download(self, dev_ptr, shape):
        out = np.zeros(shape)
        if out.nbytes:
            _lib.check(self._L.tg_memcpy_d2h(self._device, out.ctypes.data, dev_ptr, out.nbytes))
        return out

    def rollout_device(self, n_steps, dt, U_dev=None, K_dev=None, X_dev=None, max_iterations=200):
        """Asynchronous n_steps-step rollout entirely on the device (one kernel launch)."""
        self.refresh()
        _lib.check(self._L.tg_batch_rollout(self._h, int(n_steps), float(dt), U_dev, K_dev, X_dev,
                                            int(max_iterations)))

    def rollout(self, n_steps, dt, U=None, K=None, max_iterations=200):
        """Convenience: upload U [B][N][nu] / K [B][N][nk], roll out, download X [B][N+1][nX].  `dt` is a scalar or one step
        size per step (non-uniform time base)."""
        self.refresh()
        dt, dts = self._dt_argument(dt, n_steps)
        if dts is not None:
            saved = getattr(self, "_step_sizes", None)
            self.set_step_sizes(dts)
            try:
                return self.rollout(n_steps, dt, U, K, max_iterations)
            finally:
                self.set_step_sizes(*(saved if saved is not None else (None,)))
        B = self._batch
        U_dev = self.device_array(_lib.as_f64(U, (B, n_steps, self.nu))) if self.nu else None
        K_dev = self.device_array(_lib.as_f64(K, (B, n_steps, self.nk))) if self.nk else None
        X_dev = self.device_empty(B * (n_steps + 1) * self.nX)
        self.rollout_device(n_steps, dt, U_dev, K_dev, X_dev, max_iterations)
        self.synchronize()
        X = self.download(X_dev, (B, n_steps + 1, self.nX))
        for p in (U_dev, K_dev, X_dev):
            if p:
                self._L.tg_device_free(self._device, p)
                self._owned_dev.remove(p)
        return X

    def rollout_closed_loop(self, n_steps, dt, Kproj, bX, bU, group_size=1, max_iterations=200):
        """Projection-operator rollout: U_k = bU_k - Kproj_k (X_k - bX_k) evaluated in the kernel.
        Kproj [groups][N][nU][nX], bX [B][N+1][nX], bU [B][N][nU] (host arrays); returns (X, U).  `dt`: scalar or one step
        size per step."""
        self.refresh()
        dt, dts = self._dt_argument(dt, n_steps)
        if dts is not None:
            saved = getattr(self, "_step_sizes", None)
            self.set_step_sizes(dts)
            try:
                return self.rollout_closed_loop(n_steps, dt, Kproj, bX, bU, group_size, max_iterations)
            finally:
                self.set_step_sizes(*(saved if saved is not None else (None,)))
        B, nX, nU = self._batch, self.nX, self.nU
        groups = (B + group_size - 1) // group_size
        K_dev = self.device_array(_lib.as_f64(Kproj, (groups, n_steps, nU, nX)))
        bX_dev = self.device_array(_lib.as_f64(bX, (B, n_steps + 1, nX)))
        bU_dev = self.device_array(_lib.as_f64(bU, (B, n_steps, nU)))
        X_dev = self.device_empty(B * (n_steps + 1) * nX)
        U_dev = self.device_empty(B * n_steps * nU)
        _lib.check(self._L.tg_batch_rollout_closed_loop(self._h, int(n_steps), float(dt), K_dev, int(group_size),
                                                        bX_dev, bU_dev, X_dev, U_dev, int(max_iterations)))
        self.synchronize()
        X = self.download(X_dev, (B, n_steps + 1, nX))
        U = self.download(U_dev, (B, n_steps, nU))
        for p in (K_dev, bX_dev, bU_dev, X_dev, U_dev):
            self._L.tg_device_free(self._device, p)
            self._owned_dev.remove(p)
        return X, U

    def dynamics(self, Q, dQ, U=None, ddQk=None):
        """Continuous dynamics of B states at once (the reference's System.f() / System.lambda_(), system.py:951-1024):
        Q, dQ [B][nq]; U [B][nu]; ddQk [B][nk] accelerations of the kinematic configs (zeros when omitted).
        Returns (ddq [B][nd], lambda [B][nc], status [B]).  The integrator state is left alone."""
        self.refresh()
        B = self._batch
        Q = _lib.as_f64(np.broadcast_to(np.asarray(Q, dtype=float), (B, self.nq)), (B, self.nq))
        dQ = _lib.as_f64(np.broadcast_to(np.asarray(dQ, dtype=float), (B, self.nq)), (B, self.nq))
        U = np.zeros((B, self.nu)) if U is None else _lib.as_f64(np.broadcast_to(np.asarray(U, dtype=float), (B, self.nu)), (B, self.nu))
        K = np.zeros((B, self.nk)) if ddQk is None else _lib.as_f64(np.broadcast_to(np.asarray(ddQk, dtype=float), (B, self.nk)), (B, self.nk))
        ddq, lam = np.zeros((B, self.nd)), np.zeros((B, self.nc))
        status = np.zeros(B, dtype=np.int32)
        _lib.check(self._L.tg_batch_dynamics(self._h, _lib.ptr(Q), _lib.ptr(dQ), _lib.ptr(U), _lib.ptr(K),
                                             _lib.ptr(ddq), _lib.ptr(lam), status.ctypes.data))
        return ddq, lam, status

    def _seeds(self, seeds, most):
        """(seed1,) or (seed1, seed2) -> contiguous int32 [B] arrays: the input variable each trajectory's direction follows, numbered
        q [nq] | dq [nq] | ddq_k [nk] | u [nu] (forward-mode kernels, csrc/dual.hpp)."""
        if not 1 <= len(seeds) <= most:
            raise ValueError("between one and %d direction arrays" % most)
        out = [np.ascontiguousarray(np.broadcast_to(np.asarray(s_, dtype=np.int32), (self._batch,))) for s_ in seeds]
        return out

    def lagrangian(self, Q, dQ, seeds=None):
        """First and second derivatives of the Lagrangian of B states: dict with L_dq, L_ddq [B][nq] and L_dqdq,
        L_ddqdq (velocity config = row), L_ddqddq [B][nq][nq] (System.L_dq() ... L_ddqddq(), system.py:852-925).
        seeds = (s1,) or (s1, s2): the exact derivative of every entry along input variable s1[b] (and s2[b]) instead --
        the third- and fourth-order derivatives System.L_dqdqdq() ... L_ddqddqdqdq() (system.py:869-949)."""
        self.refresh()
        B = self._batch
        Q = _lib.as_f64(np.broadcast_to(np.asarray(Q, dtype=float), (B, self.nq)), (B, self.nq))
        dQ = _lib.as_f64(np.broadcast_to(np.asarray(dQ, dtype=float), (B, self.nq)), (B, self.nq))
        o1, o2 = np.zeros((B, 2, self.nq)), np.zeros((B, 3, self.nq, self.nq))
        if seeds is None:
            _lib.check(self._L.tg_batch_lagrangian(self._h, _lib.ptr(Q), _lib.ptr(dQ), _lib.ptr(o1), _lib.ptr(o2)))
        else:
            sd = self._seeds(seeds, 2)
            _lib.check(self._L.tg_batch_lagrangian_forward(self._h, _lib.ptr(Q), _lib.ptr(dQ), sd[0].ctypes.data,
                                                           sd[1].ctypes.data if len(sd) > 1 else None, _lib.ptr(o1), _lib.ptr(o2)))
        return {"L_dq": o1[:, 0], "L_ddq": o1[:, 1], "L_dqdq": o2[:, 0], "L_ddqdq": o2[:, 1], "L_ddqddq": o2[:, 2]}

    @property
    def predictor(self):
        """Initial guess of the rollouts' Newton iteration: "reference" (q2 <- previous q2, the reference's semantics
        and iteration counts) or "extrapolate" (q2 + (q2 - q1): same trajectory to solver tolerance, fewer iterations)."""
        return getattr(self, "_predictor", "reference")

    @predictor.setter
    def predictor(self, mode):
        modes = {"reference": 0, "extrapolate": 1}
        if mode not in modes:
            raise ValueError("predictor must be one of %r" % sorted(modes))
        _lib.check(self._L.tg_batch_set_predictor(self._h, modes[mode]))
        self._predictor = mode

    def energy(self, Q, dQ):
        """Kinetic and potential energy of B states: [B][2] = (T, V); the reference's System.L() is T - V and
        System.total_energy() T + V (system.py:844-850)."""
        self.refresh()
        B = self._batch
        Q = _lib.as_f64(np.broadcast_to(np.asarray(Q, dtype=float), (B, self.nq)), (B, self.nq))
        dQ = _lib.as_f64(np.broadcast_to(np.asarray(dQ, dtype=float), (B, self.nq)), (B, self.nq))
        out = np.zeros((B, 2))
        _lib.check(self._L.tg_batch_energy(self._h, _lib.ptr(Q), _lib.ptr(dQ), _lib.ptr(out)))
        return out

    DYN_D1_NAMES = ("f_dq", "f_ddq", "f_dddk", "f_du", "lambda_dq", "lambda_ddq", "lambda_dddk", "lambda_du")

    def dynamics_deriv1(self, Q, dQ, U=None, ddQk=None, seeds=None):
        """First derivatives of the continuous dynamics of B states at once (System.f_dq() ... lambda_du() of the
        reference, system.py:961-1044).  Returns ({name: [B][output][derivative variable]}, status [B]).
        seeds = (s,): the exact derivative of every entry along input variable s[b] instead (the first-derivative kernel on dual
        numbers): the second derivatives System.f_dqdq() ... lambda_dudu() (system.py:982-1078), one variable per trajectory."""
        self.refresh()
        B = self._batch
        Q = _lib.as_f64(np.broadcast_to(np.asarray(Q, dtype=float), (B, self.nq)), (B, self.nq))
        dQ = _lib.as_f64(np.broadcast_to(np.asarray(dQ, dtype=float), (B, self.nq)), (B, self.nq))
        U = np.zeros((B, self.nu)) if U is None else _lib.as_f64(np.broadcast_to(np.asarray(U, dtype=float), (B, self.nu)), (B, self.nu))
        K = np.zeros((B, self.nk)) if ddQk is None else _lib.as_f64(np.broadcast_to(np.asarray(ddQk, dtype=float), (B, self.nk)), (B, self.nk))
        rows = (self.nq, self.nq, self.nk, self.nu)
        outs = [np.zeros((B, rows[g & 3], self.nd if g < 4 else self.nc)) for g in range(8)]
        status = np.zeros(B, dtype=np.int32)
        if seeds is None:
            _lib.check(self._L.tg_batch_dynamics_deriv1(self._h, _lib.ptr(Q), _lib.ptr(dQ), _lib.ptr(U), _lib.ptr(K),
                                                        *([_lib.ptr(o) for o in outs] + [status.ctypes.data])))
        else:
            sd = self._seeds(seeds, 1)
            _lib.check(self._L.tg_batch_dynamics_deriv1_forward(self._h, _lib.ptr(Q), _lib.ptr(dQ), _lib.ptr(U), _lib.ptr(K), sd[0].ctypes.data,
                                                                *([_lib.ptr(o) for o in outs] + [status.ctypes.data])))
        return dict((n, np.swapaxes(o, 1, 2)) for n, o in zip(self.DYN_D1_NAMES, outs)), status

    def snapshot(self):
        """Save the integrator state on the device (replayed by restore())."""
        _lib.check(self._L.tg_batch_snapshot(self._h))

    def restore(self):
        _lib.check(self._L.tg_batch_restore(self._h))

    def status(self):
        iters = np.zeros(self._batch, dtype=np.int32)
        status = np.zeros(self._batch, dtype=np.int32)
        _lib.check(self._L.tg_batch_status(self._h, iters.ctypes.data, status.ctypes.data))
        return iters, status

    def solver_fallbacks(self):
        """Per trajectory: Newton systems of the last rollout / step launch that the structured solve of the specialised kernel handed
        to the pivoting solver (a failed pivot guard; results are unaffected).  tg_batch_solver_fallbacks."""
        out = np.zeros(self.batch, dtype=np.int32)
        _lib.check(self._L.tg_batch_solver_fallbacks(self._h, out.ctypes.data_as(_lib._c_ip)))
        return out

    def synchronize(self):
        _lib.check(self._L.tg_batch_synchronize(self._h))

    def set_stream(self, hip_stream):
        _lib.check(self._L.tg_batch_set_stream(self._h, hip_stream))

    def timing(self, reset=True):
        """(number of kernel launches, summed HIP-event duration in ms) since the last reset."""
        import ctypes
        n = ctypes.c_int32()
        ms = ctypes.c_double()
        _lib.check(self._L.tg_batch_timing(self._h, 1 if reset else 0, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value


class MidpointVI(object):
    """Drop-in for ``trep.MidpointVI`` (one trajectory) on top of a batch of one."""

    def __init__(self, system, tolerance=1e-10, num_threads=None, device=0, specialize="auto"):
        # num_threads: accepted and ignored (the reference's pthread pool, midpointvi.c:9-293,
        # is replaced by GPU parallelism).  specialize: as in BatchMidpointVI (not a reference argument).
        self._system = system
        self._device = device
        self._tolerance = tolerance
        self._specialize = specialize
        self._b = None
        self._cache = 0
        self._rebuild()
        system.add_structure_changed_func(self._structure_updated)

    def _structure_updated(self):
        self._stale = True

    def _rebuild(self):
        if self._b is not None:
            self._b.close()
        self._b = BatchMidpointVI(self._system, 1, self._tolerance, self._device, specialize=self._specialize)
        self._stale = False

    def _batch(self):
        if self._stale:
            # BatchMidpointVI.refresh() rebuilds the device schedule and, when the sizes are unchanged (a parameter
            # write such as `gravity.gravity = ...`), carries t, q, p, u, lambda over like the reference does
            if self._b.refresh():
                self._cache = 0
            self._stale = False
        return self._b

    def __repr__(self):
        return "<MidpointVI t1=%f t2=%f nd=%d nk=%d nc=%d nu=%d>" % (self.t1, self.t2, self.nd, self.nk, self.nc, self.nu)

    system = property(lambda self: self._system)
    nq = property(lambda self: self._batch().nq)
    nd = property(lambda self: self._batch().nd)
    nk = property(lambda self: self._batch().nk)
    nu = property(lambda self: self._batch().nu)
    nc = property(lambda self: self._batch().nc)

    @property
    def tolerance(self):
        return self._tolerance

    @tolerance.setter
    def tolerance(self, value):
        self._tolerance = value
        self._batch().tolerance = value

    def _vec(name):
        def getter(self):
            return getattr(self._batch(), name)[0].copy()

        def setter(self, value):
            self._cache = 0
            setattr(self._batch(), name, np.asarray(value, dtype=float)[None, :])
        return property(getter, setter)

    q1 = _vec("q1")
    q2 = _vec("q2")
    p1 = _vec("p1")
    p2 = _vec("p2")
    u1 = _vec("u1")
    lambda1 = _vec("lambda1")
    del _vec

    @property
    def t1(self):
        return self._batch().times()[0]

    @t1.setter
    def t1(self, t):
        self._cache = 0
        self._batch().set_times(t, self.t2)

    @property
    def t2(self):
        return self._batch().times()[1]

    @t2.setter
    def t2(self, t):
        self._cache = 0
        self._batch().set_times(self.t1, t)

    @property
    def v2(self):
        """Discrete kinematic velocity (q2k - q1k)/(t2 - t1) (midpointvi.py:325-332)."""
        t1, t2 = self._batch().times()
        if t2 != t1:
            return ((self.q2 - self.q1) / (t2 - t1))[self.nd:]
        return None

    # -- derivative accessors (midpointvi.py:337-371, 474-506, 604-640): Config / Input / Constraint
    #    objects or None (= whole axis) select entries; the result is output-major like the reference.
    def _calc_deriv1(self):
        if self._cache & 2:
            return
        if not (self._cache & 1):
            raise Exception("Integrator has not solved of the next time step yet.")
        b = self._batch()
        b.calc_deriv1()
        self._d1 = dict((n, b.deriv1(n)[0]) for n in b.D1_NAMES)
        self._cache |= 2

    def deriv2_contract(self, z):
        """HZ [R][R] for one contraction vector z (nX); see BatchMidpointVI.deriv2_contract."""
        if not (self._cache & 1):
            raise Exception("Integrator has not solved of the next time step yet.")
        return self._batch().deriv2_contract(np.asarray(z, dtype=float)[None, :])[0]

    # -- full second-derivative tensors (midpointvi.py:373-469, 508-600): assembled from 2*nd unit-vector
    #    contractions run as ONE batched launch on a helper batch holding copies of the solved step.
    def _calc_deriv2(self):
        if self._cache & 4:
            return
        if not (self._cache & 1):
            raise Exception("Integrator has not solved of the next time step yet.")
        b = self._batch()
        nd, nq, nu, nk = b.nd, b.nq, b.nu, b.nk
        nc = b.nc
        if getattr(self, "_b2", None) is None or self._b2_version != self._system._structure_version:
            self._b2 = BatchMidpointVI(self._system, max(2 * nd + nc, 1), self._tolerance, self._device, specialize=self._specialize)
            self._b2_version = self._system._structure_version
        h = self._b2
        h.set_times(*b.times())
        for name in ("q1", "q2", "p1", "p2", "u1", "lambda1"):
            setattr(h, name, getattr(b, name)[0])
        # one unit contraction per output: nd for q2, nd for p2, nc for lambda1
        Z = np.zeros((2 * nd + nc, b.nX))
        ZL = np.zeros((2 * nd + nc, nc))
        Z[np.arange(nd), np.arange(nd)] = 1.0
        Z[nd + np.arange(nd), nq + np.arange(nd)] = 1.0
        ZL[2 * nd + np.arange(nc), np.arange(nc)] = 1.0
        HZ = h.deriv2_contract(Z, ZL if nc else None)
        off = {"dq1": (0, nq), "dp1": (nq, nd), "du1": (nq + nd, nu), "dk2": (nq + nd + nu, nk)}
        self._d2 = {}
        names = ["dq1", "dp1", "du1", "dk2"]
        for ia, a in enumerate(names):
            for bname in names[ia:]:
                (oa, na), (ob, nb) = off[a], off[bname]
                blk = HZ[:, oa:oa + na, ob:ob + nb]
                self._d2["q2_" + a + bname] = np.ascontiguousarray(np.moveaxis(blk[:nd], 0, 2))
                self._d2["p2_" + a + bname] = np.ascontiguousarray(np.moveaxis(blk[nd:2 * nd], 0, 2))
                self._d2["lambda1_" + a + bname] = np.ascontiguousarray(np.moveaxis(blk[2 * nd:], 0, 2))
        self._cache |= 4

    def _d2_accessor(name, kinds):
        def accessor(self, out=None, var1=None, var2=None):
            self._calc_deriv2()
            out_kind = "c" if name.startswith("lambda1") else "d"
            return self._d2[name][self._index(var1, kinds[0]), self._index(var2, kinds[1]), self._index(out, out_kind)].copy()
        accessor.__name__ = name
        return accessor

    for _pre in ("q2", "p2", "lambda1"):
        for _pair, _kinds in (("dq1dq1", "qq"), ("dq1dp1", "qd"), ("dq1du1", "qu"), ("dq1dk2", "qk"), ("dp1dp1", "dd"),
                              ("dp1du1", "du"), ("dp1dk2", "dk"), ("du1du1", "uu"), ("du1dk2", "uk"), ("dk2dk2", "kk")):
            locals()["%s_%s" % (_pre, _pair)] = _d2_accessor("%s_%s" % (_pre, _pair), _kinds)
    del _d2_accessor, _pre, _pair, _kinds

    @staticmethod
    def _index(obj, kind):
        if obj is None:
            return slice(None)
        if kind == "k":
            assert obj.kinematic
            return obj.k_index
        if kind == "d":
            assert not obj.kinematic
        return obj.index

    def _d1_accessor(name, out_kind, var_kind):
        def accessor(self, out=None, var=None):
            self._calc_deriv1()
            return self._d1[name][self._index(var, var_kind), self._index(out, out_kind)].T.copy()
        accessor.__name__ = name
        return accessor

    q2_dq1 = _d1_accessor("q2_dq1", "d", "q")
    q2_dp1 = _d1_accessor("q2_dp1", "d", "d")
    q2_du1 = _d1_accessor("q2_du1", "d", "u")
    q2_dk2 = _d1_accessor("q2_dk2", "d", "k")
    p2_dq1 = _d1_accessor("p2_dq1", "d", "q")
    p2_dp1 = _d1_accessor("p2_dp1", "d", "d")
    p2_du1 = _d1_accessor("p2_du1", "d", "u")
    p2_dk2 = _d1_accessor("p2_dk2", "d", "k")
    lambda1_dq1 = _d1_accessor("l1_dq1", "c", "q")
    lambda1_dp1 = _d1_accessor("l1_dp1", "c", "d")
    lambda1_du1 = _d1_accessor("l1_du1", "c", "u")
    lambda1_dk2 = _d1_accessor("l1_dk2", "c", "k")
    del _d1_accessor

    def initialize_from_state(self, t1, q1, p1, lambda1=None):
        self._cache = 0
        self._batch().initialize_from_state(t1, np.asarray(q1, dtype=float)[None, :], np.asarray(p1, dtype=float)[None, :],
                                            None if lambda1 is None else np.asarray(lambda1, dtype=float)[None, :])

    def initialize_from_configs(self, t0, q0, t1, q1, lambda1=None):
        self._cache = 0
        self._batch().initialize_from_configs(t0, np.asarray(q0, dtype=float)[None, :], t1, np.asarray(q1, dtype=float)[None, :],
                                              None if lambda1 is None else np.asarray(lambda1, dtype=float)[None, :])

    def calc_p2(self):
        self._batch().calc_p2()

    def calc_f(self):
        return self._batch().calc_f()[0]

    def set_midpoint(self):
        """Put the System object at the midpoint of the current step: q = (q1 + q2) / 2, dq = (q2 - q1) / (t2 - t1), u = u1,
        t = (t1 + t2) / 2 (midpointvi.c:430-455).  Host-side state of the model object only; the device state is untouched."""
        sys_ = self._system
        sys_.t = 0.5 * (self.t1 + self.t2)
        sys_.q = 0.5 * (np.asarray(self.q1) + np.asarray(self.q2))
        sys_.dq = (np.asarray(self.q2) - np.asarray(self.q1)) / (self.t2 - self.t1)
        if self.nu:
            sys_.u = self.u1

    def discrete_fm2(self):
        """The discrete forcing of the step, fm2_i = (t2 - t1) sum_forces f(q_i) at the midpoint state, for every dynamic
        config (midpointvi.c:478-482, 2710-2728).  Evaluated through the host-side force queries (Force.f)."""
        self.set_midpoint()
        dt = self.t2 - self.t1
        return np.array([dt * sum(f.f(q) for f in self._system.forces) for q in self._system.dyn_configs])

    def step(self, t2, u1=tuple(), k2=tuple(), max_iterations=200, q2_hint=None, lambda1_hint=None):
        """Advance to t2; returns the Newton iteration count, raises ConvergenceError like the
        reference (midpointvi.py:174-201; midpointvi.c:715-718; singular -> :198-201)."""
        u1 = np.array(u1, dtype=float)
        k2 = np.array(k2, dtype=float)
        assert u1.shape == (self.nu,)
        assert k2.shape == (self.nk,)
        self._cache = 0     # the reference's setters zero the cache before the solve (midpointvi.py:188-197, 250-323)
        iters, status = self._batch().step(t2, u1[None, :], k2[None, :], max_iterations, q2_hint, lambda1_hint)
        if status[0] == _lib.NOT_CONVERGED:
            raise ConvergenceError("failed to converge after %d iterations" % (max_iterations + 1))
        if status[0] == _lib.SINGULAR:
            raise ConvergenceError("Singular derivative of DEL at t=%s" % t2)
        self._cache = 1
        return int(iters[0])
