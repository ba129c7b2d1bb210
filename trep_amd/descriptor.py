"""Flatten a ``System`` into the ``tg_system_desc`` tables of include/trep_amd.h.

The integer tables written here are the "frame indexing / tree topology" that
must match the reference bit-for-bit (SURVEY.md §8 a-T): frame order, parent,
transform kind, driving config, cache_size, cache_index, config_gen, k_index,
per-config mass lists and the global mass list, as the reference derives them in
trep/system.py:672-771 and trep/frame.py:658-691.
"""
import ctypes

import numpy as np

from . import frame as _frame
from .dynamics import Gravity, ConfigSpring, NonlinearConfigSpring, LinearSpring, Damping, ConfigForce, HybridWrench, SpatialWrench, BodyWrench, LinearDamper, Distance, PointToPoint1D, PointOnPlane

_I32 = ctypes.POINTER(ctypes.c_int32)
_F64 = ctypes.POINTER(ctypes.c_double)

_INT_SCALARS = ["n_frames", "n_configs", "n_dyn", "n_kin", "n_inputs", "n_constraints", "n_masses",
                "n_gravity", "n_damping", "n_config_forces"]
_ARRAYS = [
    ("frame_transform", _I32), ("frame_parent", _I32), ("frame_config", _I32),
    ("frame_value", _F64), ("frame_lg", _F64), ("frame_inertia", _F64),
    ("frame_cache_size", _I32), ("frame_cache_index", _I32),
    ("config_kinematic", _I32), ("config_k_index", _I32), ("config_gen", _I32),
    ("config_masses_off", _I32), ("config_masses", _I32), ("masses", _I32),
    ("gravity", _F64), ("damping", _F64),
    ("config_force_config", _I32), ("config_force_input", _I32),
    ("constraint_type", _I32), ("constraint_frame1", _I32), ("constraint_frame2", _I32),
    ("constraint_config", _I32), ("constraint_component", _I32),
    ("constraint_distance", _F64), ("constraint_tolerance", _F64),
]


_TAIL = [("n_config_springs", None), ("config_spring_config", _I32), ("config_spring_k", _F64), ("config_spring_q0", _F64),
         ("n_linear_springs", None), ("linear_spring_frame1", _I32), ("linear_spring_frame2", _I32),
         ("linear_spring_k", _F64), ("linear_spring_x0", _F64), ("constraint_normal", _F64),
         ("n_hybrid_wrenches", None), ("hybrid_wrench_frame", _I32), ("hybrid_wrench_input", _I32), ("hybrid_wrench_const", _F64), ("hybrid_wrench_kind", _I32),
         ("n_linear_dampers", None), ("linear_damper_frame1", _I32), ("linear_damper_frame2", _I32), ("linear_damper_c", _F64),
         ("n_nonlinear_springs", None), ("nonlinear_spring_config", _I32), ("nonlinear_spring_m", _F64), ("nonlinear_spring_b", _F64),
         ("nonlinear_spring_first", _I32), ("nonlinear_spring_pieces", _F64)]   # in struct order (include/trep_amd.h)
_TAIL_SCALARS = [n for n, t in _TAIL if t is None]
_TAIL_ARRAYS = [(n, t) for n, t in _TAIL if t is not None]


class SystemDescStruct(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in _INT_SCALARS] + _ARRAYS +
                [(n, ctypes.c_int32 if t is None else t) for n, t in _TAIL])


class SystemDesc(object):
    """numpy tables + a ctypes view of them (keeps the arrays alive)."""

    def __init__(self, tables):
        self.tables = tables
        self.struct = SystemDescStruct()
        for n in _INT_SCALARS + _TAIL_SCALARS:
            setattr(self.struct, n, int(tables[n]))
        for n, ptype in _ARRAYS + _TAIL_ARRAYS:
            arr = tables[n]
            want = np.int32 if ptype is _I32 else np.float64
            assert arr.dtype == want and arr.flags["C_CONTIGUOUS"], n
            if arr.size == 0:  # keep a valid (never dereferenced) pointer
                arr = np.zeros(1, dtype=want)
                tables["_pad_" + n] = arr
            setattr(self.struct, n, arr.ctypes.data_as(ptype))

    def byref(self):
        return ctypes.byref(self.struct)

    def __getattr__(self, name):
        try:
            return self.__dict__["tables"][name]
        except KeyError:
            raise AttributeError(name)


def flatten(system):
    """Return the SystemDesc of a trep_amd.System."""
    system._sync()
    frames = system.frames
    configs = system.configs
    nq = len(configs)
    fidx = {id(f): i for i, f in enumerate(frames)}
    cidx = {id(c): i for i, c in enumerate(configs)}

    t = {}
    t["n_frames"] = len(frames)
    t["n_configs"] = nq
    t["n_dyn"] = system.nQd
    t["n_kin"] = system.nQk
    t["n_inputs"] = system.nu
    t["n_constraints"] = system.nc
    t["n_masses"] = len(system.masses)

    nf = len(frames)
    t["frame_transform"] = np.array([f.transform_type.code for f in frames], dtype=np.int32)
    t["frame_parent"] = np.array([-1 if f.parent is None else fidx[id(f.parent)] for f in frames], dtype=np.int32)
    t["frame_config"] = np.array([-1 if f.config is None else cidx[id(f.config)] for f in frames], dtype=np.int32)
    t["frame_value"] = np.array([0.0 if f.config is not None else f._value for f in frames], dtype=np.float64)
    lg = np.zeros((nf, 12), dtype=np.float64)
    for i, f in enumerate(frames):
        m = f._lg_const if f.transform_type is _frame.CONST_SE3 else np.eye(4)
        lg[i] = np.asarray(m)[:3, :].reshape(12)
    t["frame_lg"] = lg.reshape(-1)
    t["frame_inertia"] = np.array([[f._mass, f._Ixx, f._Iyy, f._Izz] for f in frames],
                                  dtype=np.float64).reshape(-1)
    t["frame_cache_size"] = np.array([f._cache_size for f in frames], dtype=np.int32)
    ci = -np.ones((nf, nq + 1), dtype=np.int32)
    for i, f in enumerate(frames):
        for j in range(f._cache_size):
            ci[i, j] = cidx[id(f._cache_index[j])]
    t["frame_cache_index"] = ci.reshape(-1)

    t["config_kinematic"] = np.array([1 if c.kinematic else 0 for c in configs], dtype=np.int32)
    t["config_k_index"] = np.array([c._k_index for c in configs], dtype=np.int32)
    t["config_gen"] = np.array([c._config_gen for c in configs], dtype=np.int32)
    off = [0]
    cm = []
    for c in configs:
        cm += [fidx[id(f)] for f in c._masses]
        off.append(len(cm))
    t["config_masses_off"] = np.array(off, dtype=np.int32)
    t["config_masses"] = np.array(cm, dtype=np.int32)
    t["masses"] = np.array([fidx[id(f)] for f in system.masses], dtype=np.int32)

    grav, damp, cf_c, cf_u = [], [], [], []
    cs_c, cs_k, cs_q0 = [], [], []
    ns_c, ns_m, ns_b, ns_first, ns_rows = [], [], [], [0], []
    ls_f1, ls_f2, ls_k, ls_x0 = [], [], [], []
    hw_f, hw_in, hw_c, hw_kind = [], [], [], []
    ld_f1, ld_f2, ld_c = [], [], []
    for pot in system.potentials:
        if isinstance(pot, Gravity):
            grav.append(list(pot._gravity))
        elif isinstance(pot, LinearSpring):
            ls_f1.append(fidx[id(pot.frame1)])
            ls_f2.append(fidx[id(pot.frame2)])
            ls_k.append(pot.k)
            ls_x0.append(pot.x0)
        elif isinstance(pot, ConfigSpring):
            cs_c.append(cidx[id(pot.config)])
            cs_k.append(pot.k)
            cs_q0.append(pot.q0)
        elif isinstance(pot, NonlinearConfigSpring):
            ns_c.append(cidx[id(pot.config)])
            ns_m.append(float(pot.m))
            ns_b.append(float(pot.b))
            xp, co = pot.spline.x_points, pot.spline.coefficients
            ns_rows += [[xp[k]] + list(co[k]) for k in range(len(co))]
            ns_first.append(len(ns_rows))
        else:
            raise NotImplementedError("potential %r is outside the device path's scope" % (pot,))
    for force in system.forces:
        if isinstance(force, Damping):
            damp.append(force.coefficient_array())
        elif isinstance(force, HybridWrench):   # and its subclasses SpatialWrench, BodyWrench
            hw_kind.append(2 if isinstance(force, BodyWrench) else (1 if isinstance(force, SpatialWrench) else 0))
            hw_f.append(fidx[id(force.frame)])
            hw_in += [-1 if v is None else v._index for v in force._wrench_vars]
            hw_c += [float(c) for c in force._wrench_cons]
        elif isinstance(force, LinearDamper):
            ld_f1.append(fidx[id(force.frame1)])
            ld_f2.append(fidx[id(force.frame2)])
            ld_c.append(float(force.c))
        elif isinstance(force, ConfigForce):
            cf_c.append(cidx[id(force.config)])
            cf_u.append(force.finput._index)
        else:
            raise NotImplementedError("force %r is outside the device path's scope" % (force,))
    t["n_gravity"] = len(grav)
    t["n_damping"] = len(damp)
    t["n_config_forces"] = len(cf_c)
    t["gravity"] = np.array(grav, dtype=np.float64).reshape(-1)
    t["damping"] = np.array(damp, dtype=np.float64).reshape(-1)
    t["config_force_config"] = np.array(cf_c, dtype=np.int32)
    t["config_force_input"] = np.array(cf_u, dtype=np.int32)
    t["n_config_springs"] = len(cs_c)
    t["config_spring_config"] = np.array(cs_c, dtype=np.int32)
    t["config_spring_k"] = np.array(cs_k, dtype=np.float64)
    t["config_spring_q0"] = np.array(cs_q0, dtype=np.float64)
    t["n_linear_springs"] = len(ls_k)
    t["linear_spring_frame1"] = np.array(ls_f1, dtype=np.int32)
    t["linear_spring_frame2"] = np.array(ls_f2, dtype=np.int32)
    t["linear_spring_k"] = np.array(ls_k, dtype=np.float64)
    t["linear_spring_x0"] = np.array(ls_x0, dtype=np.float64)
    t["n_hybrid_wrenches"] = len(hw_f)
    t["hybrid_wrench_frame"] = np.array(hw_f, dtype=np.int32)
    t["hybrid_wrench_input"] = np.array(hw_in, dtype=np.int32)
    t["hybrid_wrench_const"] = np.array(hw_c, dtype=np.float64)
    t["hybrid_wrench_kind"] = np.array(hw_kind, dtype=np.int32)
    t["n_nonlinear_springs"] = len(ns_c)
    t["nonlinear_spring_config"] = np.array(ns_c, dtype=np.int32)
    t["nonlinear_spring_m"] = np.array(ns_m, dtype=np.float64)
    t["nonlinear_spring_b"] = np.array(ns_b, dtype=np.float64)
    t["nonlinear_spring_first"] = np.array(ns_first, dtype=np.int32)
    t["nonlinear_spring_pieces"] = np.array(ns_rows, dtype=np.float64).reshape(-1)
    t["n_linear_dampers"] = len(ld_c)
    t["linear_damper_frame1"] = np.array(ld_f1, dtype=np.int32)
    t["linear_damper_frame2"] = np.array(ld_f2, dtype=np.int32)
    t["linear_damper_c"] = np.array(ld_c, dtype=np.float64)

    ctype, cf1, cf2, ccfg, ccomp, cdist, ctol = [], [], [], [], [], [], []
    cnormal = []
    for con in system.constraints:
        cnormal.append([0.0, 0.0, 0.0])
        if isinstance(con, Distance):
            ctype.append(0)
            ccfg.append(-1 if con.config is None else cidx[id(con.config)])
            ccomp.append(0)
            cdist.append(con._distance)
        elif isinstance(con, PointOnPlane):
            ctype.append(2)
            ccfg.append(-1)
            ccomp.append(0)
            cdist.append(0.0)
            cnormal[-1] = [float(x) for x in con.normal]
        elif isinstance(con, PointToPoint1D):
            ctype.append(1)
            ccfg.append(-1)
            ccomp.append(con.component)
            cdist.append(0.0)
        else:
            raise NotImplementedError("constraint %r is outside the device path's scope" % (con,))
        cf1.append(fidx[id(con.frame1)])
        cf2.append(fidx[id(con.frame2)])
        ctol.append(con.tolerance)
    t["constraint_type"] = np.array(ctype, dtype=np.int32)
    t["constraint_frame1"] = np.array(cf1, dtype=np.int32)
    t["constraint_frame2"] = np.array(cf2, dtype=np.int32)
    t["constraint_config"] = np.array(ccfg, dtype=np.int32)
    t["constraint_component"] = np.array(ccomp, dtype=np.int32)
    t["constraint_distance"] = np.array(cdist, dtype=np.float64)
    t["constraint_tolerance"] = np.array(ctol, dtype=np.float64)
    t["constraint_normal"] = np.array(cnormal, dtype=np.float64).reshape(-1)
    return SystemDesc(t)
