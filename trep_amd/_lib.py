"""ctypes binding of libtrepamd.so (the HIP library; C ABI in include/trep_amd.h).

There is deliberately no fallback: if the shared library is missing or no HIP
device is visible, the integrator raises instead of computing anything on the CPU.
"""
import ctypes
import os

import numpy as np

from .descriptor import SystemDescStruct

_HERE = os.path.dirname(os.path.abspath(__file__))
# TREPAMD_LIB may point at a diagnostic build of the same HIP library (e.g. libtrepamd_prof.so)
LIB_PATH = os.environ.get("TREPAMD_LIB") or os.path.join(_HERE, "libtrepamd.so")
_LIB = None

_c_dp = ctypes.POINTER(ctypes.c_double)
_c_ip = ctypes.POINTER(ctypes.c_int32)

# field ids of include/trep_amd.h
F_Q1, F_Q2, F_P1, F_P2, F_U1, F_LAMBDA1 = 0, 1, 2, 3, 4, 5
F_D1_BASE = 10  # TG_F_Q2_DQ1; order q2_d{q1,p1,u1,k2}, p2_d*, l1_d*
OK, NOT_CONVERGED, SINGULAR = 0, 1, 2



class LqProblem(ctypes.Structure):
    """tg_lq_problem of include/trep_amd.h (device pointers as integers)."""
    _fields_ = [("n_problems", ctypes.c_int32), ("horizon", ctypes.c_int32), ("nX", ctypes.c_int32), ("nU", ctypes.c_int32),
                ("select_dev", ctypes.c_void_p),
                ("A_dev", ctypes.c_void_p), ("B_dev", ctypes.c_void_p),
                ("Q_dev", ctypes.c_void_p), ("Q_seed_stride", ctypes.c_int64), ("Q_step_stride", ctypes.c_int64),
                ("Qf_dev", ctypes.c_void_p), ("Qf_seed_stride", ctypes.c_int64),
                ("R_dev", ctypes.c_void_p), ("R_seed_stride", ctypes.c_int64), ("R_step_stride", ctypes.c_int64),
                ("hz_dev", ctypes.c_void_p), ("hz_R", ctypes.c_int32), ("hz_nx", ctypes.c_int32),
                ("q_dev", ctypes.c_void_p), ("r_dev", ctypes.c_void_p),
                ("K_dev", ctypes.c_void_p), ("C_dev", ctypes.c_void_p), ("P0_dev", ctypes.c_void_p), ("b0_dev", ctypes.c_void_p),
                ("status_dev", ctypes.c_void_p), ("b_next_dev", ctypes.c_void_p),
                ("ds_nd", ctypes.c_int32), ("ds_nk", ctypes.c_int32), ("ds_nu", ctypes.c_int32),
                ("k_begin", ctypes.c_int32), ("k_end", ctypes.c_int32), ("Pt_dev", ctypes.c_void_p), ("bt_dev", ctypes.c_void_p)]


_vp, _i32, _f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_double

_SIGNATURES = {
    "tg_version": (ctypes.c_char_p, []),
    "tg_last_error": (ctypes.c_char_p, []),
    "tg_device_count": (ctypes.c_int, []),
    "tg_device_info": (ctypes.c_int, [ctypes.c_int32, _c_ip]),
    "tg_system_create": (ctypes.c_void_p, [ctypes.POINTER(SystemDescStruct)]),
    "tg_system_destroy": (None, [ctypes.c_void_p]),
    "tg_system_sizes": (ctypes.c_int, [ctypes.c_void_p, _c_ip]),
    "tg_system_info": (ctypes.c_int, [ctypes.c_void_p, _c_ip]),
    "tg_batch_create": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]),
    "tg_batch_destroy": (None, [ctypes.c_void_p]),
    "tg_batch_set_tolerance": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
    "tg_batch_set_times": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]),
    "tg_batch_get_times": (ctypes.c_int, [ctypes.c_void_p, _c_dp, _c_dp]),
    "tg_batch_set": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "tg_batch_get": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "tg_batch_field_width": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "tg_batch_calc_p2": (ctypes.c_int, [ctypes.c_void_p]),
    "tg_batch_calc_f": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "tg_batch_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                     ctypes.c_void_p]),
    "tg_batch_rollout": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    "tg_batch_rollout_closed_loop": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p,
                                                    ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_int32]),
    "tg_batch_rollout_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), _c_ip]),
    "tg_batch_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tg_batch_solver_fallbacks": (ctypes.c_int, [ctypes.c_void_p, _c_ip]),
    "tg_batch_deriv1": (ctypes.c_int, [ctypes.c_void_p]),
    "tg_batch_deriv2_contract": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tg_batch_dynamics": (ctypes.c_int, [ctypes.c_void_p] * 8),
    "tg_batch_dynamics_device": (ctypes.c_int, [ctypes.c_void_p] * 8),
    "tg_batch_dynamics_deriv1": (ctypes.c_int, [ctypes.c_void_p] * 14),
    "tg_batch_dynamics_deriv1_device": (ctypes.c_int, [ctypes.c_void_p] * 7),
    "tg_batch_energy": (ctypes.c_int, [ctypes.c_void_p] * 4),
    "tg_batch_lagrangian": (ctypes.c_int, [ctypes.c_void_p] * 5),
    "tg_batch_lagrangian_forward": (ctypes.c_int, [ctypes.c_void_p] * 7),
    "tg_batch_dynamics_deriv1_forward": (ctypes.c_int, [ctypes.c_void_p] * 15),
    "tg_batch_set_predictor": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "tg_batch_deriv2_contract_lambda": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tg_batch_snapshot": (ctypes.c_int, [ctypes.c_void_p]),
    "tg_batch_restore": (ctypes.c_int, [ctypes.c_void_p]),
    "tg_device_alloc": (ctypes.c_void_p, [ctypes.c_int32, ctypes.c_uint64]),
    "tg_device_free": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p]),
    "tg_memcpy_h2d": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "tg_memcpy_d2h": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "tg_batch_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "tg_batch_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "tg_batch_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _c_ip, _c_dp]),
    "tg_debug_solve": (ctypes.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "tg_system_newton_plan": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "tg_batch_debug_newton_solve": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "tg_batch_set_pivot_rule": (ctypes.c_int, [_vp, _i32]),
    "tg_batch_set_step_sizes": (ctypes.c_int, [_vp, _i32, _vp, _i32]),
    "tg_system_spec_header": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64]),
    "tg_system_spec_key": (ctypes.c_uint64, [ctypes.c_void_p]),
    "tg_batch_load_specialized": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "tg_batch_info": (ctypes.c_int, [ctypes.c_void_p, _c_ip]),
    "tg_batch_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    # device-side discopt primitives
    "tg_batch_set_from_trajectories": (ctypes.c_int, [_vp, _i32, _i32, _f64, _f64, _vp, _vp, _i32]),
    "tg_batch_linearize": (ctypes.c_int, [_vp, _vp, _vp]),
    "tg_batch_initialize_from_state_device": (ctypes.c_int, [_vp, _f64, _vp, ctypes.c_uint64]),
    "tg_batch_deriv2_contract_device": (ctypes.c_int, [_vp, _vp, _vp]),
    "tg_batch_deriv2_contract_device_range": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32]),
    "tg_tv_lq": (ctypes.c_int, [_i32, ctypes.POINTER(LqProblem)]),
    "tg_adjoint_sweep": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tg_tangent_rollout": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tg_quadratic_cost": (ctypes.c_int, [_i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tg_batch_rollout_closed_loop_subset": (ctypes.c_int, [_vp, _i32, _i32, _f64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32]),
    "tg_quadratic_cost_gradients": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tg_armijo_candidates": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tg_copy_rows": (ctypes.c_int, [_i32, _i32, ctypes.c_uint64, _vp, _vp, _vp, _vp]),
    "tg_device_synchronize": (ctypes.c_int, [_i32]),
    "tg_dopt_use_stream": (ctypes.c_int, [_i32, _i32]),
    "tg_dopt_lane_stream": (_vp, [_i32, _i32]),
    "tg_dopt_lane_wait": (ctypes.c_int, [_i32, _i32, _i32]),
    # multi-GPU: RCCL all-gather / scalar reductions (csrc/comm.hip)
    "tg_comm_unique_id": (ctypes.c_int, [_vp]),
    "tg_comm_create": (_vp, [_i32, _i32, _i32, _vp]),
    "tg_comm_destroy": (None, [_vp]),
    "tg_comm_info": (ctypes.c_int, [_vp, _c_ip]),
    "tg_comm_all_gather": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint64]),
    "tg_comm_wait_stream": (ctypes.c_int, [_vp, _vp]),
    "tg_comm_all_gather_after": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_uint64]),
    "tg_comm_stream_wait_comm": (ctypes.c_int, [_vp, _vp]),
    "tg_comm_synchronize": (ctypes.c_int, [_vp]),
    "tg_comm_all_reduce_host": (ctypes.c_int, [_vp, _c_dp, _i32, _i32]),
    "tg_comm_barrier": (ctypes.c_int, [_vp]),
}


class LibraryError(RuntimeError):
    pass


def lib():
    """Load libtrepamd.so (once).  Raises LibraryError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two streams that share a queue run
        # one after the other.  The device-resident optimiser keeps up to four sweeps / kernels in flight in stream lanes of their own beside
        # the batches' streams (discopt/batch_doptimizer.py): with four queues the lanes collide and its pipelined Newton step runs at the
        # speed of the unpipelined one (67 instead of 39 ms at 32 seeds, measured).  A default only -- a value the user has set is kept --,
        # and it must be in the environment before the HIP runtime initialises, i.e. before the library that links it is loaded.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        L = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = L
    return _LIB


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc):
    if rc != 0:
        raise LibraryError("libtrepamd error %d: %s" % (rc, lib().tg_last_error().decode()))


def require_device():
    n = lib().tg_device_count()
    if n <= 0:
        raise LibraryError("no HIP device visible; trep_amd has no CPU execution path")
    return n


def as_f64(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shape):
        raise ValueError("expected array of shape %r, got %r" % (tuple(shape), a.shape))
    return a


def ptr(a):
    return None if a is None or a.size == 0 else a.ctypes.data
