"""BatchDOptimizer: S independent trajectory-optimisation problems ("seeds") of one system advanced together,
device resident.

Same algorithm as ``DOptimizer`` (reference trep/discopt/doptimizer.py:207-566: projection-operator descent with
a steepest / quasi-Newton / Newton LQ model, Armijo search over closed-loop projections, fallback
newton -> quasi -> steepest) with a quadratic ``DCost`` per seed -- but every stage runs on the GPU for all
seeds at once and the trajectories, linearisations, gains and candidates never leave HBM:

  stage                                   kernel(s) (include/trep_amd.h)                      parallel axis
  --------------------------------------  --------------------------------------------------  ---------------
  DSystem.set + fdx/fdu for all (s,k)     tg_batch_set_from_trajectories, tg_batch_linearize  S*N trajectories
  projection gain (solve_tv_lqr)          tg_tv_lq                                            S workgroups
  cost, gradients                         tg_quadratic_cost, tg_quadratic_cost_gradients      S / S*(N+1)
  Newton model: adjoint + fdxdx/xu/uu(z)  tg_adjoint_sweep, tg_batch_deriv2_contract_device   S / S*N
  descent direction (solve_tv_lq + dX,dU) tg_tv_lq, tg_tangent_rollout                        S workgroups
  Armijo candidates, projection, costs    tg_armijo_candidates, tg_batch_rollout_closed_loop  S*M trajectories

Only per-seed scalars (costs, directional derivatives, statuses) come back to the host, which takes the
accept / fallback / terminate decisions exactly like the reference's ``step``.

The two backward sweeps of a step that do not depend on each other -- the projection gain (Riccati on A, B alone) and the
quasi-Newton LQ model (A, B, the cost's weights and gradients) -- are each a chain of N dependent steps on ONE workgroup
per seed.  With at most half as many seeds as the GPU has CUs they are launched side by side on two streams
(``overlap_sweeps``, tg_dopt_use_stream): in a quasi step that hides the projection gain behind the descent direction,
in a Newton step the quasi direction is already there for the seeds whose Newton model is not a descent direction
(doptimizer.py:482-494 computes it only then; the numbers are the same either way).
"""
import os
from collections import namedtuple

import numpy as np

from .. import _lib
from ..midpointvi import BatchMidpointVI

METHODS = ("steepest", "quasi", "newton")


class _DeviceArray(object):
    """A typed view of a device allocation (owned by a _DevicePool)."""

    def __init__(self, pool, shape, dtype):
        self.pool, self.shape, self.dtype = pool, tuple(int(x) for x in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = pool._alloc(max(self.nbytes, 8))

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.shape != self.shape:
            raise ValueError("expected shape %r, got %r" % (self.shape, host.shape))
        if self.nbytes:
            _lib.check(self.pool.L.tg_memcpy_h2d(self.pool.device, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def get(self):
        out = np.zeros(self.shape, dtype=self.dtype)
        if self.nbytes:
            _lib.check(self.pool.L.tg_memcpy_d2h(self.pool.device, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def get_block(self, index):
        """One leading-index block (e.g. A[s, k]) without downloading the whole array."""
        index = tuple(int(i) for i in index)
        sub = self.shape[len(index):]
        out = np.zeros(sub, dtype=self.dtype)
        flat = int(np.ravel_multi_index(index, self.shape[:len(index)])) * out.nbytes
        if out.nbytes:
            _lib.check(self.pool.L.tg_memcpy_d2h(self.pool.device, out.ctypes.data, int(self.ptr) + flat, out.nbytes))
        return out


class _DevicePool(object):
    def __init__(self, device):
        self.L = _lib.lib()
        _lib.require_device()
        self.device = device
        self._ptrs = []

    def _alloc(self, nbytes):
        p = self.L.tg_device_alloc(self.device, nbytes)
        if not p:
            raise _lib.LibraryError(self.L.tg_last_error().decode())
        self._ptrs.append(p)
        return p

    def empty(self, shape, dtype=np.float64):
        return _DeviceArray(self, shape, dtype)

    def upload(self, host, dtype=np.float64):
        host = np.ascontiguousarray(host, dtype=dtype)
        return _DeviceArray(self, host.shape, dtype).set(host)

    def close(self):
        for p in self._ptrs:
            self.L.tg_device_free(self.device, p)
        self._ptrs = []


class BatchDOptimizer(object):
    """S seeds of the same DSystem, each with its own desired trajectory (xd, ud) and shared weights Q, R, Qf."""

    step_return = namedtuple("batch_step", "done cost0 dcost0 cost1 method armijo failed")


    def __init__(self, dsys, Xd, Ud, Q, R, Qf=None, device=0, armijo_chunk=None, first_method_iterations=10,
                 predictor="reference", overlap_sweeps="auto", pipeline_newton="auto"):
        self.dsys = dsys
        ds = dsys
        Xd = np.asarray(Xd, dtype=float)
        Ud = np.asarray(Ud, dtype=float)
        self.S, self.N = Xd.shape[0], Xd.shape[1] - 1
        self.nX, self.nU = ds.nX, ds.nU
        if Xd.shape != (self.S, self.N + 1, self.nX) or Ud.shape != (self.S, self.N, self.nU):
            raise ValueError("Xd must be [S][N+1][nX] and Ud [S][N][nU]")
        if self.N != ds.kf():
            raise ValueError("the DSystem's time base has %d steps, the trajectories %d" % (ds.kf(), self.N))
        steps = np.diff(np.asarray(ds.time, dtype=float))
        self.dt, self.t0 = float(steps[0]), float(ds.time[0])
        self._steps = None if np.allclose(steps, steps[0], rtol=1e-13, atol=0.0) else steps      # non-uniform time base
        self.armijo_beta = 0.7
        self.armijo_alpha = 0.00001
        self.armijo_max_iterations = 30
        self._armijo_hint = {}
        self.descent_tolerance = 1e-6
        self.first_method_iterations = first_method_iterations
        self.first_method = "quasi"
        self.second_method = "newton"
        self.device = device
        self.L = _lib.lib()
        self.pool = pool = _DevicePool(device)
        S, N, nX, nU = self.S, self.N, self.nX, self.nU
        sysm = ds.system
        self._nxh = len(sysm.configs) + len(sysm.dyn_configs)      # leading [Q; p] part of X the curvature covers
        self._R = self._nxh + nU
        # block structure of DSystem.fdx / fdu (dsystem.py:284-317) for the LQ sweeps: TREPAMD_LQ_DENSE=1 runs them dense
        nd_, nk_ = len(sysm.dyn_configs), len(sysm.kin_configs)
        self._ds = (nd_, nk_, nU - nk_) if (2 * (nd_ + nk_) == nX and os.environ.get("TREPAMD_LQ_DENSE", "0") != "1") else (0, 0, 0)
        self._ds_checked = False          # the skipped blocks of A_k / B_k are looked at once, after the first linearisation
        # engines: horizon batch (one trajectory per (seed, step)) and candidate batch
        if armijo_chunk is None:   # fill the GPU once (256 CUs x 8 wavefronts) but never more than the search needs
            armijo_chunk = int(min(self.armijo_max_iterations, max(1, 2048 // S)))
        self.M = int(armijo_chunk)
        self.lin = BatchMidpointVI(sysm, S * N, device=device)
        self.arm = BatchMidpointVI(sysm, S * self.M, device=device)
        self.arm.predictor = predictor    # Newton start of the Armijo projections ("extrapolate": opt-in warm start)
        if self._steps is not None:       # trajectory (s, k) of the horizon batch steps by t[k+1] - t[k]; step k of a projection too
            self.lin.set_step_sizes(self._steps, by_trajectory=True)
            self.arm.set_step_sizes(self._steps)
        # device state
        self.Xd, self.Ud = pool.upload(Xd), pool.upload(Ud)
        self.Q = pool.upload(np.asarray(Q, dtype=float).reshape(nX, nX))
        self.R = pool.upload(np.asarray(R, dtype=float).reshape(nU, nU))
        self.Qf = pool.upload(np.asarray(Q if Qf is None else Qf, dtype=float).reshape(nX, nX))
        self.Ix, self.Iu = pool.upload(np.eye(nX)), pool.upload(np.eye(nU))
        self.X, self.U = pool.empty((S, N + 1, nX)), pool.empty((S, N, nU))
        self.A, self.B = pool.empty((S, N, nX, nX)), pool.empty((S, N, nX, nU))
        self.Kproj, self.K, self.C = pool.empty((S, N, nU, nX)), pool.empty((S, N, nU, nX)), pool.empty((S, N, nU))
        self.q, self.r = pool.empty((S, N + 1, nX)), pool.empty((S, N, nU))
        self.dX, self.dU = pool.empty((S, N + 1, nX)), pool.empty((S, N, nU))
        self.dcost, self.cost = pool.empty((S,)), pool.empty((S,))
        self.lq_status = pool.empty((S,), np.int32)
        # second set of direction buffers: the quasi-Newton sweep that runs beside the projection gain
        info = np.zeros(4, dtype=np.int32)
        _lib.check(self.L.tg_device_info(device, info.ctypes.data_as(_lib._c_ip)))
        self.CUS = int(info[0])      # one sweep workgroup occupies one compute unit
        self.overlap = (2 * S <= self.CUS) if overlap_sweeps == "auto" else bool(overlap_sweeps)
        if self.overlap:
            self.K2, self.C2 = pool.empty((S, N, nU, nX)), pool.empty((S, N, nU))
            self.dX2, self.dU2 = pool.empty((S, N + 1, nX)), pool.empty((S, N, nU))
            self.dcost2, self.lq_status2 = pool.empty((S,)), pool.empty((S,), np.int32)
        # pipelined Newton step (whenever the sweeps run side by side, i.e. the seeds leave half of the CUs idle): the projection sweep,
        # the z-contracted second derivatives and the Newton-model sweep run chunk by chunk of the horizon in three stream lanes, the
        # quasi-Newton sweep in a fourth (projection_quasi_and_newton_model).  TREPAMD_NEWTON_PIPELINE=0/1 overrides, TREPAMD_NEWTON_CHUNKS
        env_pipe = os.environ.get("TREPAMD_NEWTON_PIPELINE")
        if pipeline_newton == "auto":
            self.pipeline = self.overlap if env_pipe is None else (env_pipe == "1" and self.overlap)      # (128 seeds on 256 CUs still gain 5 %)
        else:
            self.pipeline = bool(pipeline_newton) and self.overlap
        self.pipeline_chunks = int(os.environ.get("TREPAMD_NEWTON_CHUNKS", "24"))
        if self.pipeline:
            self.Pc = [(pool.empty((S, nX, nX)), pool.empty((S, nX))) for _ in range(4)]      # (P, b) carried between the chunks: two per sweep
            self.lq_status3 = pool.empty((S,), np.int32)
            stream = self.L.tg_dopt_lane_stream(device, 3)
            if not stream:
                raise _lib.LibraryError(self.L.tg_last_error().decode())
            self.lin.set_stream(stream)       # the horizon batch's kernels (linearisation, second derivatives) run in lane 3
        self._newton_ready = None     # (dcost [S], failed [S]) of the pipelined Newton model of this step, or None
        self._quasi_ready = None      # (dcost [S], failed [S]) of the side-by-side quasi sweep of this step, or None
        self._adjoint_ready = False   # Z holds the adjoint of the current iterate (written by the projection sweep)
        self.fuse_adjoint = os.environ.get("TREPAMD_SEPARATE_ADJOINT") is None      # (the env switch is for A/B measurements)
        self.Z = None
        self.HZ = None
        self.bX, self.bU = pool.empty((S * self.M, N + 1, nX)), pool.empty((S * self.M, N, nU))
        self.cX, self.cU = pool.empty((S * self.M, N + 1, nX)), pool.empty((S * self.M, N, nU))
        self.ccost = pool.empty((S * self.M,))
        self.lambdas = pool.empty((self.M,))
        self.x0 = pool.empty((S, nX))
        self._sel = pool.empty((S,), np.int32)
        self._rows_a, self._rows_b = pool.empty((S,), np.int32), pool.empty((S,), np.int32)
        self._lq_failed = np.zeros(S, dtype=bool)
        self.iteration = 0

    def close(self):
        for e in (self.lin, self.arm):
            if e is not None:
                e.close()
        self.lin = self.arm = None
        self.pool.close()

    # -- trajectories --------------------------------------------------------------------------------------
    def set_trajectories(self, X, U):
        self.X.set(X)
        self.U.set(U)

    def get_trajectories(self):
        return self.X.get(), self.U.get()

    # -- stages (each one launch or a few; all seeds) ----------------------------------------------------
    def _check(self, rc):
        _lib.check(rc)

    def _select(self, seeds):
        """Device index list for a subset of the seeds (None = all)."""
        if seeds is None:
            return None, self.S
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        buf = np.zeros(self.S, dtype=np.int32)
        buf[:len(seeds)] = seeds
        self._sel.set(buf)
        return self._sel.ptr, len(seeds)

    def linearize(self):
        """A, B about the current (X, U) for every seed and step; leaves the S*N solved steps resident."""
        self.lin.refresh()
        self._adjoint_ready = False
        self._check(self.L.tg_batch_set_from_trajectories(self.lin._h, self.S, self.N, self.t0, self.dt, self.X.ptr, self.U.ptr, 200))
        _, status = self.lin.status()
        self._check(self.L.tg_batch_linearize(self.lin._h, self.A.ptr, self.B.ptr))
        if self._ds[0] and not self._ds_checked:
            self._check_ds_structure()
        # per seed: a seed with a failed DEL solve anywhere along its horizon has no linearisation (the reference raises
        # ConvergenceError out of DSystem.set for that one problem); the other seeds are unaffected
        return (status.reshape(self.S, self.N) != 0).any(axis=1)

    def _check_ds_structure(self):
        """One-off (first linearisation): the structured LQ sweep skips A's Qk rows / v columns and B's single-entry rows on the promise
        that they hold EXACT zeros (tg_lq_problem::ds_*).  The promise is the linearisation kernel's; this looks at three (seed, step)
        blocks and falls back to the dense sweep -- loudly -- if it is ever broken (TREPAMD_LQ_DENSE=1 is the manual override)."""
        nd, nk, nu = self._ds
        nq, nX = nd + nk, self.nX
        ok = True
        for s, k in {(0, 0), (self.S - 1, self.N - 1), (self.S // 2, self.N // 2)}:
            A, B = self.A.get_block((s, k)), self.B.get_block((s, k))
            rows_qk, rows_v = slice(nd, nq), slice(nq + nd, nX)
            skipA = A.copy()
            skipA[:nd, :nq + nd] = 0.0
            skipA[nq:nq + nd, :nq + nd] = 0.0                      # dense Qd and p rows over the [Q, p] columns
            skipA[rows_v, nd:nq] -= np.diag(np.diag(A[rows_v, nd:nq]))  # a v row holds one entry: its Qk column
            skipB = B.copy()
            skipB[:nd] = 0.0
            skipB[nq:nq + nd] = 0.0                                # dense Qd and p rows
            skipB[rows_qk, nu:] -= np.diag(np.diag(B[rows_qk, nu:]))    # Qk and v rows: one entry each, their rho column
            skipB[rows_v, nu:] -= np.diag(np.diag(B[rows_v, nu:]))
            ok = ok and not skipA.any() and not skipB.any()
        self._ds_checked = True
        if not ok:
            import warnings
            warnings.warn("BatchDOptimizer: A_k / B_k do not have the DSystem block structure the structured LQ sweep assumes; "
                          "using the dense sweep")
            self._ds = (0, 0, 0)

    def _lq(self, seeds, Q, Qf, R, hz, affine, K, C=None, status=None, collect=True, b_next=None, k_range=None, terminal=None, carry=None):
        """k_range = (k0, k1): sweep only the steps k1 - 1 ... k0; terminal = (P, b) device arrays at step k1 (None: the horizon's end);
        carry = (P, b) arrays that receive (P, b) at step k0 (tg_lq_problem::k_begin / k_end / Pt_dev / bt_dev / P0_dev / b0_dev)."""
        sel, n = self._select(seeds)
        if n == 0:
            return
        status = self.lq_status if status is None else status
        p = _lib.LqProblem()
        p.n_problems, p.horizon, p.nX, p.nU = n, self.N, self.nX, self.nU
        p.select_dev = sel
        p.A_dev, p.B_dev = self.A.ptr, self.B.ptr
        p.Q_dev, p.Q_seed_stride, p.Q_step_stride = Q.ptr, 0, 0
        p.Qf_dev, p.Qf_seed_stride = Qf.ptr, 0
        p.R_dev, p.R_seed_stride, p.R_step_stride = R.ptr, 0, 0
        p.hz_dev = hz.ptr if hz is not None else None
        p.hz_R, p.hz_nx = self._R, self._nxh
        p.q_dev, p.r_dev = (self.q.ptr, self.r.ptr) if affine else (None, None)
        p.K_dev, p.C_dev = K.ptr, (C.ptr if C is not None else None)
        p.P0_dev, p.b0_dev = (carry[0].ptr, carry[1].ptr) if carry is not None else (None, None)
        p.k_begin, p.k_end = (int(k_range[0]), int(k_range[1])) if k_range is not None else (0, 0)
        p.Pt_dev, p.bt_dev = (terminal[0].ptr, terminal[1].ptr) if terminal is not None else (None, None)
        p.b_next_dev = b_next.ptr if (b_next is not None and affine) else None
        p.status_dev = status.ptr
        p.ds_nd, p.ds_nk, p.ds_nu = self._ds
        import ctypes
        self._check(self.L.tg_tv_lq(self.device, ctypes.byref(p)))
        if collect:
            self._lq_collect(seeds, status)

    def _lq_collect(self, seeds, status, into=None):
        st = status.get()               # TG_SINGULAR: the nU x nU matrix of a Riccati step of that seed has no usable pivot
        idx = np.arange(self.S) if seeds is None else np.asarray(seeds, dtype=np.int64)      # status is indexed by seed
        (self._lq_failed if into is None else into)[idx[st[idx] != 0]] = True

    def _ensure_newton_buffers(self):
        if self.Z is None:
            self.Z = self.pool.empty((self.S, self.N, self.nX))
            self.HZ = self.pool.empty((self.S, self.N, self._R, self._R))

    def projection_gain(self, with_adjoint=False, collect=True):
        """Kproj = solve_tv_lqr(A, B, I, I) (doptimizer.py:272-287).  with_adjoint (needs q, r: gradients_and_cost first): the same
        sweep run with the cost gradients as its affine terms -- its vector recursion b_k = q_k - K_k' r_k + (A_k - B_k K_k)' b_{k+1}
        IS the adjoint z of the Newton model (doptimizer.py:340-343), so Z[s][k] = z_{k+1} comes out of the projection sweep and the
        separate backward sweep (tg_adjoint_sweep) is not needed."""
        if with_adjoint:
            self._ensure_newton_buffers()
            self._lq(None, self.Ix, self.Ix, self.Iu, None, True, self.Kproj, None, collect=collect, b_next=self.Z)
        else:
            self._lq(None, self.Ix, self.Ix, self.Iu, None, False, self.Kproj, collect=collect)
        self._adjoint_ready = bool(with_adjoint)

    def projection_gain_and_quasi_direction(self, with_adjoint=False):
        """The projection gain and, beside it on a second stream, the quasi-Newton direction of EVERY seed into the second
        set of direction buffers (needs q, r: call gradients_and_cost first).  Leaves (dcost, failed) in _quasi_ready."""
        S, N, nX, nU = self.S, self.N, self.nX, self.nU
        # fork: everything the two sweeps read (A, B from the linearisation's stream, q, r from the cost kernels) is complete before the
        # lanes start -- an explicit device synchronisation (tens of microseconds against sweeps of tens of milliseconds) instead of
        # relying on how blocking streams order themselves against the default stream; the join below is explicit as well
        self._check(self.L.tg_device_synchronize(self.device))
        try:
            self._check(self.L.tg_dopt_use_stream(self.device, 1))
            self.projection_gain(with_adjoint, collect=False)
            self._check(self.L.tg_dopt_use_stream(self.device, 2))
            self._lq(None, self.Q, self.Qf, self.R, None, True, self.K2, self.C2, status=self.lq_status2, collect=False)
            self._check(self.L.tg_tangent_rollout(self.device, S, N, nX, nU, None, self.A.ptr, self.B.ptr, self.K2.ptr, self.C2.ptr,
                                                  self.q.ptr, self.r.ptr, self.dX2.ptr, self.dU2.ptr, self.dcost2.ptr))
        finally:
            self._check(self.L.tg_dopt_use_stream(self.device, 0))
        self._check(self.L.tg_device_synchronize(self.device))
        self._lq_collect(None, self.lq_status)
        failed = np.zeros(S, dtype=bool)
        self._lq_collect(None, self.lq_status2, into=failed)
        self._quasi_ready = (self.dcost2.get(), failed)

    def _chunks(self):
        """The horizon cut into pipeline_chunks ranges of steps, last range first (the sweeps run backwards)."""
        n = max(1, min(self.pipeline_chunks, self.N // 16))
        edges = np.linspace(0, self.N, n + 1).astype(int)
        return [(int(edges[i]), int(edges[i + 1])) for i in range(n)][::-1]

    def projection_quasi_and_newton_model(self):
        """Everything between the linearisation and the line search of a Newton step, for EVERY seed, as a pipeline over chunks of the
        horizon (needs q, r: gradients_and_cost first):

            lane 1  projection gain with the adjoint z riding on it        chunk c: steps [k0, k1)
            lane 3  second derivatives contracted with z                   chunk c after lane 1's chunk c        (the horizon batch's stream)
            lane 4  LQ sweep of the Newton model, then its tangent rollout  chunk c after lane 3's chunk c
            lane 2  quasi-Newton LQ sweep + tangent rollout (the fallback direction), whole horizon

        A sweep continues a chunk from the (P, b) the previous chunk left (tg_lq_problem::Pt_dev), so gains and directions are bit for bit
        those of the three sweeps run one after the other (doptimizer.py:319-402 runs them in that order); what changes is that the
        seed-count-independent serial time of a Newton step is one sweep (plus a chunk of each of the others) instead of three.
        Leaves the Newton direction in dX, dU (_newton_ready) and the quasi direction in the second buffers (_quasi_ready)."""
        S, N, nX, nU = self.S, self.N, self.nX, self.nU
        self._ensure_newton_buffers()
        L, dev = self.L, self.device
        self._check(L.tg_device_synchronize(dev))
        chunks = self._chunks()
        try:
            self._check(L.tg_dopt_use_stream(dev, 2))
            self._lq(None, self.Q, self.Qf, self.R, None, True, self.K2, self.C2, status=self.lq_status2, collect=False)
            self._check(L.tg_tangent_rollout(dev, S, N, nX, nU, None, self.A.ptr, self.B.ptr, self.K2.ptr, self.C2.ptr,
                                             self.q.ptr, self.r.ptr, self.dX2.ptr, self.dU2.ptr, self.dcost2.ptr))
            for c, (k0, k1) in enumerate(chunks):
                first = c == 0
                self._check(L.tg_dopt_use_stream(dev, 1))
                self._lq(None, self.Ix, self.Ix, self.Iu, None, True, self.Kproj, None, collect=False, b_next=self.Z, k_range=(k0, k1),
                         terminal=None if first else self.Pc[(c - 1) % 2], carry=self.Pc[c % 2])
                self._check(L.tg_dopt_lane_wait(dev, 3, 1))
                self._check(L.tg_batch_deriv2_contract_device_range(self.lin._h, self.Z.ptr, self.HZ.ptr, N, k0, k1))
                self._check(L.tg_dopt_lane_wait(dev, 4, 3))
                self._check(L.tg_dopt_use_stream(dev, 4))
                self._lq(None, self.Q, self.Qf, self.R, self.HZ, True, self.K, self.C, status=self.lq_status3, collect=False, k_range=(k0, k1),
                         terminal=None if first else self.Pc[2 + (c - 1) % 2], carry=self.Pc[2 + c % 2])
            self._check(L.tg_tangent_rollout(dev, S, N, nX, nU, None, self.A.ptr, self.B.ptr, self.K.ptr, self.C.ptr,
                                             self.q.ptr, self.r.ptr, self.dX.ptr, self.dU.ptr, self.dcost.ptr))
        finally:
            self._check(L.tg_dopt_use_stream(dev, 0))
        self._check(L.tg_device_synchronize(dev))
        self._adjoint_ready = True
        self._lq_collect(None, self.lq_status)
        failed = np.zeros(S, dtype=bool)
        self._lq_collect(None, self.lq_status2, into=failed)
        self._quasi_ready = (self.dcost2.get(), failed)
        failed3 = np.zeros(S, dtype=bool)
        self._lq_collect(None, self.lq_status3, into=failed3)
        self._newton_ready = (self.dcost.get(), failed3)

    def _take_newton_direction(self, seeds):
        """dcost of `seeds` for the Newton direction the pipeline left in dX, dU (rows of other seeds are overwritten by whoever takes them)."""
        dc, failed = self._newton_ready
        idx = np.arange(self.S) if seeds is None else np.asarray(seeds, dtype=np.int64)
        self._lq_failed[idx[failed[idx]]] = True
        return dc[idx]

    def _take_quasi_direction(self, seeds):
        """dX, dU of `seeds` (None = all) <- the quasi direction computed beside the projection gain; returns their dcost."""
        dc, failed = self._quasi_ready
        S, N, nX, nU = self.S, self.N, self.nX, self.nU
        if seeds is None:
            self.dX, self.dX2 = self.dX2, self.dX
            self.dU, self.dU2 = self.dU2, self.dU
            self._quasi_ready = None          # (the buffers now hold whatever dX, dU held)
            idx = np.arange(S)
        else:
            idx = np.asarray(seeds, dtype=np.int64)
            sel, n = self._select(seeds)
            self._check(self.L.tg_copy_rows(self.device, n, (N + 1) * nX, sel, sel, self.dX2.ptr, self.dX.ptr))
            self._check(self.L.tg_copy_rows(self.device, n, N * nU, sel, sel, self.dU2.ptr, self.dU.ptr))
        self._lq_failed[idx[failed[idx]]] = True
        return dc[idx]

    def gradients_and_cost(self):
        S, N, nX, nU = self.S, self.N, self.nX, self.nU
        self._check(self.L.tg_quadratic_cost_gradients(self.device, S, N, nX, nU, None, self.X.ptr, self.U.ptr, self.Xd.ptr, self.Ud.ptr,
                                                       self.Q.ptr, self.R.ptr, self.Qf.ptr, self.q.ptr, self.r.ptr))
        self._check(self.L.tg_quadratic_cost(self.device, S, 1, None, N, nX, nU, self.X.ptr, self.U.ptr, self.Xd.ptr, self.Ud.ptr,
                                             self.Q.ptr, self.R.ptr, self.Qf.ptr, self.cost.ptr))
        return self.cost.get()

    def newton_curvature(self, seeds):
        """HZ[s][k] = second derivatives of step k contracted with the adjoint z_{k+1} (doptimizer.py:319-345)."""
        self._ensure_newton_buffers()
        if not self._adjoint_ready:       # (else: Z came out of the projection sweep)
            sel, n = self._select(seeds)
            self._check(self.L.tg_adjoint_sweep(self.device, n, self.N, self.nX, self.nU, sel, self.A.ptr, self.B.ptr, self.Kproj.ptr,
                                                self.q.ptr, self.r.ptr, self.Z.ptr))
        self._check(self.L.tg_batch_deriv2_contract_device(self.lin._h, self.Z.ptr, self.HZ.ptr))

    def descent_direction(self, seeds, method):
        """K, C from the LQ model of `method`, then dX, dU and the directional derivative for `seeds`."""
        if method == "steepest":
            self._lq(seeds, self.Ix, self.Ix, self.Iu, None, True, self.K, self.C)
        elif method == "quasi":
            self._lq(seeds, self.Q, self.Qf, self.R, None, True, self.K, self.C)
        elif method == "newton":
            self.newton_curvature(seeds)
            self._lq(seeds, self.Q, self.Qf, self.R, self.HZ, True, self.K, self.C)
        else:
            raise ValueError("Invalid descent direction method: %r" % method)
        sel, n = self._select(seeds)
        self._check(self.L.tg_tangent_rollout(self.device, n, self.N, self.nX, self.nU, sel, self.A.ptr, self.B.ptr, self.K.ptr, self.C.ptr,
                                              self.q.ptr, self.r.ptr, self.dX.ptr, self.dU.ptr, self.dcost.ptr))

    def armijo_chunk(self, m0, seeds=None, count=None):
        """Project the candidates lambda = beta^m, m0 <= m < m0 + count, of `seeds` (default: all seeds, count = M);
        returns (costs [n][count], ok [n][count]).  Row i*count + j of the candidate buffers belongs to seeds[i]."""
        S, N, nX, nU = self.S, self.N, self.nX, self.nU
        sel, n = self._select(seeds)
        M = self.M if count is None else int(count)
        assert n * M <= S * self.M
        if M > self.lambdas.shape[0]:
            self.lambdas = self.pool.empty((M,))
        lam = np.zeros(self.lambdas.shape[0])
        lam[:M] = self.armijo_beta ** np.arange(m0, m0 + M, dtype=float)
        self.lambdas.set(lam)
        rows = n * M
        self._check(self.L.tg_armijo_candidates(self.device, n, M, N, nX, nU, sel, self.lambdas.ptr, self.X.ptr, self.U.ptr,
                                                self.dX.ptr, self.dU.ptr, self.bX.ptr, self.bU.ptr))
        self._check(self.L.tg_batch_initialize_from_state_device(self.arm._h, self.t0, self.bX.ptr, (N + 1) * nX))
        self._check(self.L.tg_batch_rollout_closed_loop_subset(self.arm._h, rows, N, self.dt, self.Kproj.ptr, M, sel, self.bX.ptr,
                                                               self.bU.ptr, self.cX.ptr, self.cU.ptr, 200))
        self._check(self.L.tg_quadratic_cost(self.device, rows, M, sel, N, nX, nU, self.cX.ptr, self.cU.ptr, self.Xd.ptr, self.Ud.ptr,
                                             self.Q.ptr, self.R.ptr, self.Qf.ptr, self.ccost.ptr))
        _, status = self.arm.status()
        return self.ccost.get()[:rows].reshape(n, M), (status[:rows] == 0).reshape(n, M)

    def accept(self, seeds, rows):
        """X[s], U[s] <- row rows[i] of the candidate buffers for s = seeds[i]."""
        n = len(seeds)
        if n == 0:
            return
        S, N, nX, nU = self.S, self.N, self.nX, self.nU
        a = np.zeros(S, dtype=np.int32)
        b = np.zeros(S, dtype=np.int32)
        a[:n] = seeds
        b[:n] = rows
        self._rows_a.set(a)
        self._rows_b.set(b)
        self._check(self.L.tg_copy_rows(self.device, n, (N + 1) * nX, self._rows_a.ptr, self._rows_b.ptr, self.cX.ptr, self.X.ptr))
        self._check(self.L.tg_copy_rows(self.device, n, N * nU, self._rows_a.ptr, self._rows_b.ptr, self.cU.ptr, self.U.ptr))

    # -- iteration -----------------------------------------------------------------------------------------
    def select_method(self, iteration):
        return self.first_method if iteration < self.first_method_iterations else self.second_method

    @staticmethod
    def _fallback(method):
        if method == "newton":
            return "quasi"
        if method == "quasi":
            return "steepest"
        raise Exception("Derivative of cost is positive for steepest descent.")

    def step(self, method="steepest", active=None):
        """One DOptimizer.step for every active seed (doptimizer.py:462-506).  `method` is a name or a
        per-seed list.  Returns arrays over all seeds (inactive seeds: done=True, costs NaN)."""
        S = self.S
        self.lin.refresh()      # parameter writes on the system since the engines were built (see BatchMidpointVI.refresh)
        self.arm.refresh()
        active = np.ones(S, dtype=bool) if active is None else np.asarray(active, dtype=bool).copy()
        methods = np.array([method] * S if isinstance(method, str) else list(method), dtype=object)
        self._lq_failed[:] = False
        broken = self.linearize() & active
        self._quasi_ready = None
        cost0 = self.gradients_and_cost()                       # (q, r do not depend on the projection gain; the sweeps below use them)
        newton = self.fuse_adjoint and any(m == "newton" for m in methods[active])    # then the projection sweep also carries the adjoint
        self._newton_ready = None
        if self.pipeline and newton:
            self.projection_quasi_and_newton_model()
        elif self.overlap and any(m in ("quasi", "newton") for m in methods[active]):
            self.projection_gain_and_quasi_direction(with_adjoint=newton)
        else:
            self.projection_gain(with_adjoint=newton)
        broken |= self._lq_failed & active
        active &= ~broken               # their step ends here (flagged failed below); everything is per seed from now on
        dcost0 = np.full(S, np.nan)
        pending = active.copy()
        while pending.any():
            for name in METHODS:
                seeds = np.nonzero(pending & (methods == name))[0]
                if len(seeds) == 0:
                    continue
                if name == "quasi" and self._quasi_ready is not None:
                    dcost0[seeds] = self._take_quasi_direction(None if len(seeds) == S else seeds)
                    continue
                if name == "newton" and self._newton_ready is not None:
                    dcost0[seeds] = self._take_newton_direction(None if len(seeds) == S else seeds)
                    continue
                self.descent_direction(None if len(seeds) == S else seeds, name)
                dc = self.dcost.get()
                dcost0[seeds] = dc[seeds]
            lost = pending & (self._lq_failed | ~np.isfinite(dcost0))     # singular LQ model: no direction for that seed
            broken |= lost
            active &= ~lost
            pending &= ~lost
            bad = pending & (dcost0 > 0)
            pending = bad
            for s in np.nonzero(bad)[0]:
                methods[s] = self._fallback(methods[s])
        done = ~active | (np.abs(dcost0) < self.descent_tolerance)
        cost1 = np.where(active, cost0, np.nan)
        armijo = np.full(S, -1)
        search = active & ~done
        searched = search.copy()
        m0 = 0
        # How far the first round speculates: M candidates per seed fill the GPU, but a projection at one wave per SIMD takes 30 us per DEL
        # step and at two 38 -- so when the last step with the same method accepted early everywhere (quasi-Newton steps: m <= 3 on the
        # puppet), the first round only goes twice as far as that step needed.  Which candidate a seed accepts does not depend on the
        # batching: the candidates are still tried in order, a seed that needs more gets them in the next round.
        hints = [self._armijo_hint.get(m) for m in set(methods[search])]
        first = self.M if (not hints or any(h is None for h in hints)) else max(4, 2 * (max(hints) + 1))
        while search.any() and m0 < self.armijo_max_iterations:
            # first round: every seed, M candidates each (one full wave of the GPU); later rounds: only the seeds
            # still searching, with as many of their remaining candidates as fit into the candidate batch
            if m0 == 0:
                seeds, count = np.arange(S), min(self.M, self.armijo_max_iterations, first)
                costs, ok = self.armijo_chunk(0, None, count)
            else:
                seeds = np.nonzero(search)[0]
                count = int(min(self.armijo_max_iterations - m0, max(1, (S * self.M) // len(seeds))))
                costs, ok = self.armijo_chunk(m0, seeds, count)
            lam = self.armijo_beta ** np.arange(m0, m0 + count)
            seeds = np.asarray(seeds)
            # first candidate of every searching seed that converged and satisfies the sufficient-decrease test (doptimizer.py:436-459)
            with np.errstate(invalid="ignore"):
                good = ok[:, :count] & (costs[:, :count] < cost0[seeds, None] + self.armijo_alpha * lam[None, :] * dcost0[seeds, None])
            good &= search[seeds][:, None]
            hit = good.any(axis=1)
            j = good.argmax(axis=1)
            rows_i = np.nonzero(hit)[0]
            acc_seeds = seeds[rows_i]
            acc_rows = rows_i * costs.shape[1] + j[rows_i]
            cost1[acc_seeds] = costs[rows_i, j[rows_i]]
            armijo[acc_seeds] = m0 + j[rows_i]
            search[acc_seeds] = False
            self.accept(list(acc_seeds), list(acc_rows))
            m0 += count
        # a seed whose search is exhausted is where the reference raises ConvergenceError("Armijo Failed to
        # Converge") (doptimizer.py:456-459); here it is flagged and left unchanged, the other seeds carry on
        failed = search | broken
        for m in set(methods[searched]):          # the deepest accepted candidate of this step, per method: the next step's speculation depth
            took = armijo[searched & (methods == m) & (armijo >= 0)]
            self._armijo_hint[m] = int(took.max()) if len(took) and not (search & (methods == m)).any() else None
        self.iteration += 1
        shown = active | broken
        return self.step_return(done | failed, np.where(shown, cost0, np.nan), dcost0, np.where(broken, cost0, cost1), list(methods), armijo, failed)

    def optimize(self, max_steps=50):
        """Runs every seed until |dcost| < descent_tolerance, an Armijo failure, or max_steps;
        returns (converged [S], X, U)."""
        active = np.ones(self.S, dtype=bool)
        failed = np.zeros(self.S, dtype=bool)
        for i in range(max_steps):
            r = self.step(self.select_method(i), active)
            active &= ~r.done
            failed |= r.failed
            if not active.any():
                break
        X, U = self.get_trajectories()
        return ~active & ~failed, X, U
