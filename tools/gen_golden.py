#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ from the REAL reference.

Runs only in the build container: imports the Python-3 working copy of the
reference that tools/build_reference.py puts in /tmp/trep_ref, builds each
BASELINE system there with the same builder code this package uses
(trep_amd.systems, pointed at the reference's API), runs the reference's own
MidpointVI / DSystem and records inputs and outputs.  The .npz files hold data
only (numbers and names), never reference source.

Recorded per system:
  * topology tables (frame order/names, parents, transforms, configs, config_gen,
    cache_index, masses) for the bit-exact topology test;
  * free-running rollouts (q2, p2, lambda1, Newton iterations per step) from seeded
    synthetic initial conditions and inputs -- every step is also a teacher-forced
    case, because (q, p, lambda) at step k are the complete inputs of step k+1;
  * first-derivative arrays at several steps; second-derivative tensors (full for
    the small systems, selected + z-contracted for the puppet);
  * DSystem captures: set(X[k],U[k],k,xk_hint) -> f, fdx, fdu, fdxdx(z), fdxdu(z), fdudu(z).
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/tmp/trep_ref")

import trep  # noqa: E402  (the reference, Python-3 working copy)
import trep.puppets  # noqa: E402
import trep.discopt  # noqa: E402
from trep_amd import systems  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
DT = 0.01
D1 = ["q2_dq1", "q2_dp1", "q2_du1", "q2_dk2", "p2_dq1", "p2_dp1", "p2_du1", "p2_dk2",
      "l1_dq1", "l1_dp1", "l1_du1", "l1_dk2"]
PAIRS = ["dq1dq1", "dq1dp1", "dq1du1", "dq1dk2", "dp1dp1", "dp1du1", "dp1dk2", "du1du1", "du1dk2", "dk2dk2"]
CODES = {"WORLD": 0, "TX": 1, "TY": 2, "TZ": 3, "RX": 4, "RY": 5, "RZ": 6, "CONST_SE3": 7}


def topology(system):
    frames, configs = system.frames, system.configs
    fidx = {id(f): i for i, f in enumerate(frames)}
    cidx = {id(c): i for i, c in enumerate(configs)}
    t = {}
    t["topo_frame_names"] = np.array([str(f.name) for f in frames])
    t["topo_config_names"] = np.array([str(c.name) for c in configs])
    t["topo_frame_transform"] = np.array([CODES[str(f.transform_type)] for f in frames], dtype=np.int32)
    t["topo_frame_parent"] = np.array([-1 if f.parent is None else fidx[id(f.parent)] for f in frames], dtype=np.int32)
    t["topo_frame_config"] = np.array([-1 if f.config is None else cidx[id(f.config)] for f in frames], dtype=np.int32)
    t["topo_frame_cache_size"] = np.array([f._cache_size for f in frames], dtype=np.int32)
    t["topo_frame_cache_index"] = np.array(
        [[-1 if c is None else cidx[id(c)] for c in f._cache_index] for f in frames], dtype=np.int32).reshape(-1)
    t["topo_config_kinematic"] = np.array([1 if c.kinematic else 0 for c in configs], dtype=np.int32)
    t["topo_config_gen"] = np.array([c._config_gen for c in configs], dtype=np.int32)
    t["topo_config_k_index"] = np.array([c.k_index for c in configs], dtype=np.int32)
    t["topo_masses"] = np.array([fidx[id(f)] for f in system.masses], dtype=np.int32)
    off, cm = [0], []
    for c in configs:
        cm += [fidx[id(f)] for f in c.masses]
        off.append(len(cm))
    t["topo_config_masses_off"] = np.array(off, dtype=np.int32)
    t["topo_config_masses"] = np.array(cm, dtype=np.int32)
    t["topo_sizes"] = np.array([system.nQ, system.nQd, system.nQk, system.nu, system.nc, len(frames)], dtype=np.int32)
    return t


def rollout(system, q0, U, K, n_steps, deriv_steps=(), deriv2_full=False, deriv2_select=(), second_order=True):
    """Free-running reference rollout from initialize_from_configs(0, q0, DT, q0)."""
    mvi = trep.MidpointVI(system, num_threads=1)
    mvi.initialize_from_configs(0.0, q0, DT, q0)
    nd = mvi.nd
    Q, P, LAM, IT = [mvi.q2], [mvi.p2], [mvi.lambda1], []
    extra = {}
    for k in range(n_steps):
        it = mvi.step(mvi.t2 + DT, U[k], K[k])
        Q.append(mvi.q2); P.append(mvi.p2); LAM.append(mvi.lambda1); IT.append(it)
        if (k + 1) in deriv_steps:
            if second_order:
                mvi._calc_deriv2()
            else:
                mvi._calc_deriv1()
            for n in D1:
                extra["d1_%d_%s" % (k + 1, n)] = getattr(mvi, "_" + n).copy()
            for pr in PAIRS:
                for pre in ("q2_", "p2_", "l1_"):
                    name = pre + pr
                    if second_order and (deriv2_full or name in deriv2_select):
                        extra["d2_%d_%s" % (k + 1, name)] = getattr(mvi, "_" + name).copy()
            extra["f_%d" % (k + 1)] = mvi.calc_f()
    return dict(Q=np.array(Q), P=np.array(P), LAM=np.array(LAM).reshape(n_steps + 1, -1),
                IT=np.array(IT, dtype=np.int32), **extra)


def dsystem_captures(system, Q, P, U, K, ks, seed, second_order=True):
    """DSystem.set(X[k],U[k],k,xk_hint=X[k+1]) -> f, A, B and z-contracted second-order terms."""
    n = len(Q)
    t = DT * np.arange(n)
    mvi = trep.MidpointVI(system, num_threads=1)
    dsys = trep.discopt.DSystem(mvi, t)
    nk, nd = system.nQk, system.nQd
    V = np.zeros((n, nk))
    V[1:] = (Q[1:, nd:] - Q[:-1, nd:]) / DT
    X, Uin = dsys.build_trajectory(Q, P, V, U[:n - 1], K[:n - 1])
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((2, dsys.nX))
    out = {"ds_k": np.array(ks, dtype=np.int32), "ds_Z": Z, "ds_X": X, "ds_U": Uin}
    for k in ks:
        dsys.set(X[k], Uin[k], k, xk_hint=X[k + 1])
        out["ds_%d_f" % k] = dsys.f()
        out["ds_%d_A" % k] = dsys.fdx()
        out["ds_%d_B" % k] = dsys.fdu()
        out["ds_%d_lambda" % k] = mvi.lambda1
        for zi in range(2 if second_order else 0):
            out["ds_%d_fdxdx_%d" % (k, zi)] = dsys.fdxdx(Z[zi])
            out["ds_%d_fdxdu_%d" % (k, zi)] = dsys.fdxdu(Z[zi])
            out["ds_%d_fdudu_%d" % (k, zi)] = dsys.fdudu(Z[zi])
    return out


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f kB)" % (path, os.path.getsize(path) / 1e3))


def gen_pendulum(links, n_steps, name):
    system = systems.pendulum(links, api=trep)
    q0 = system.q
    U = np.zeros((n_steps, 0)); K = np.zeros((n_steps, 0))
    r = rollout(system, q0, U, K, n_steps, deriv_steps=(1, 50, n_steps), deriv2_full=True)
    ds = dsystem_captures(system, r["Q"][:60], r["P"][:60], U, K, (0, 10, 50), seed=11)
    save(name, q0=q0, U=U, K=K, dt=DT, **topology(system), **r, **ds)


def gen_known_answer():
    """The reference's own known-answer case (examples/papers/tase2012/pend-single-step.py:10-41):
    1-DOF pendulum m=l=1, g=9.8, q=0.2, p=0.5, u=0.8, dt=0.1."""
    s = trep.System()
    s.import_frames([trep.rx('theta', name='pend_angle'), [trep.tz(-1.0, name='pend_mass', mass=1.0)]])
    trep.potentials.Gravity(s, (0, 0, -9.8))
    trep.forces.ConfigForce(s, 'theta', 'theta-torque')
    mvi = trep.MidpointVI(s, num_threads=1)
    mvi.initialize_from_state(0.0, np.array([0.2]), np.array([0.5]))
    it = mvi.step(0.1, np.array([0.8]))
    mvi._calc_deriv2()
    out = dict(q2=mvi.q2, p2=mvi.p2, iterations=np.array([it]))
    for n in ("q2_dq1", "q2_dp1", "q2_du1", "p2_dq1", "p2_dp1", "p2_du1"):
        out[n] = getattr(mvi, "_" + n).copy()
    for n in ("q2_dq1dq1", "p2_dq1dq1", "q2_du1du1", "q2_dq1du1", "p2_dq1du1", "q2_dp1dp1"):
        out[n] = getattr(mvi, "_" + n).copy()
    save("known_answer_pendulum", **out)


def gen_cart():
    system = systems.pend_on_cart(api=trep)
    rng = np.random.default_rng(20250 + 2)
    B, N = 4, 200
    x0 = rng.uniform(-1, 1, size=4096)[:B]
    th0 = rng.uniform(-np.pi, np.pi, size=4096)[:B]
    Uall = rng.standard_normal((4096, N, 1))[:B] * 2.0
    K = np.zeros((N, 0))
    arrays = dict(dt=DT, **topology(system))
    for b in range(B):
        q0 = np.array([x0[b], th0[b]])
        r = rollout(system, q0, Uall[b], K, N, deriv_steps=(1, 50, N) if b == 0 else (), deriv2_full=True)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = q0
        arrays["b%d_U" % b] = Uall[b]
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], Uall[0], K, (0, 10, 100), seed=12)
    save("pend_on_cart", **arrays, **ds)


def gen_scissor():
    system = systems.scissor_lift(4, api=trep)
    rng = np.random.default_rng(20250 + 5)
    B, N = 2, 200
    th = rng.uniform(0.03 * np.pi, 0.12 * np.pi, size=4096)[:B]
    U = np.zeros((N, 0)); K = np.zeros((N, 0))
    arrays = dict(dt=DT, theta0=th, **topology(system))
    for b in range(B):
        q0 = systems.scissor_q(system, th[b])
        r = rollout(system, q0, U, K, N, deriv_steps=(1, 50, N) if b == 0 else (), deriv2_full=True)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = q0
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], U, K, (0, 10, 100), seed=13)
    save("scissor4", **arrays, **ds)


def gen_puppet():
    system = systems.puppet(api=trep)
    B, N = 2, 200
    Q0 = systems.puppet_initial_conditions(system, B)
    nd = system.nQd
    Kall = systems.puppet_string_schedule(system, Q0[:, nd:], N, DT)
    U = np.zeros((N, 0))
    arrays = dict(dt=DT, **topology(system))
    select = ("q2_dq1dq1", "p2_dq1dk2", "l1_dk2dk2", "q2_dp1dp1", "p2_dk2dk2")
    for b in range(B):
        r = rollout(system, Q0[b], U, Kall[b], N, deriv_steps=(1, 100) if b == 0 else (),
                    deriv2_select=select)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = Q0[b]
        arrays["b%d_K" % b] = Kall[b]
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], U, Kall[0], (0, 10, 100), seed=14)
    # the base-pose first step quoted in SURVEY.md Appendix A
    system.q = 0.0
    system.q = systems.PUPPET_BASE_POSE
    system.project_string_controls()
    q0 = system.q
    mvi = trep.MidpointVI(system, num_threads=1)
    mvi.initialize_from_configs(0.0, q0, DT, q0)
    it = mvi.step(2 * DT, (), q0[nd:])
    arrays.update(base_q0=q0, base_q2=mvi.q2, base_p2=mvi.p2, base_lambda1=mvi.lambda1, base_it=np.array([it]))
    save("puppet40", **arrays, **ds)




def gen_puppet_basic():
    """examples/puppet-basic.py: the starting guess of the script and seeded perturbations of it, each made
    consistent with the six string constraints by the reference's own System.satisfy_constraints (SLSQP; the
    build does not re-implement it, the consistent poses are fixture data), then free-running rollouts."""
    system = systems.puppet_basic(api=trep)
    rng = np.random.default_rng(20250 + 6)
    n_ic, N = 16, 200
    joints = [c.name for c in system.configs if c.name not in ('TorsoX', 'TorsoY', 'TorsoZ')]
    ics = []
    for b in range(n_ic):
        system.q = 0.0
        system.q = systems.PUPPET_BASIC_POSE
        if b:
            for name in joints:
                c = system.get_config(name)
                c.q = c.q + rng.uniform(-0.05, 0.05)
        system.satisfy_constraints()
        assert max(abs(c.h()) for c in system.constraints) < 1e-8
        ics.append(system.q.copy())
    ics = np.array(ics)
    U = np.zeros((N, 0)); K = np.zeros((N, 0))
    arrays = dict(dt=DT, ic_set=ics, **topology(system))
    select = ("q2_dq1dq1", "p2_dq1dp1", "l1_dq1dq1", "q2_dp1dp1", "p2_dp1dp1")
    for b in range(2):
        r = rollout(system, ics[b], U, K, N, deriv_steps=(1, 100) if b == 0 else (), deriv2_select=select)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = ics[b]
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], U, K, (0, 10, 100), seed=15)
    save("puppet_basic", **arrays, **ds)


def gen_spring_arm():
    """Synthetic system with ConfigSpring potentials: free-running rollouts under a
    random torque and a moving base, full derivative tensors, DSystem captures."""
    system = systems.spring_arm(api=trep)
    rng = np.random.default_rng(20250 + 7)
    B, N = 2, 200
    arrays = dict(dt=DT, **topology(system))
    for b in range(B):
        q0 = np.concatenate([rng.uniform(-0.8, 0.8, size=3), [0.2 * b]])
        U = rng.standard_normal((N, 1))
        K = (0.2 * b + 0.3 * np.sin(2.0 * DT * np.arange(1, N + 1)))[:, None]
        r = rollout(system, q0, U, K, N, deriv_steps=(1, 50, N) if b == 0 else (), deriv2_full=True)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = q0
        arrays["b%d_U" % b] = U
        arrays["b%d_K" % b] = K
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], arrays["b0_U"], arrays["b0_K"], (0, 10, 100), seed=16)
    save("spring_arm", **arrays, **ds)


def gen_nonlinear_spring_arm():
    """Synthetic system with NonlinearConfigSpring potentials (nonlinear_config_spring.c, spline.py): free-running rollouts
    under a random torque and a moving base that leave the splines' knot ranges, full derivative tensors, DSystem captures;
    also the spline tables themselves (x points, coefficients) of the three curves."""
    system = systems.nonlinear_spring_arm(api=trep)
    rng = np.random.default_rng(20250 + 27)
    B, N = 2, 200
    arrays = dict(dt=DT, **topology(system))
    for i, pot in enumerate(p for p in system.potentials if hasattr(p, "spline")):
        arrays["spline%d_x" % i] = pot.spline.x_points
        arrays["spline%d_y" % i] = pot.spline.y_points
        arrays["spline%d_c" % i] = pot.spline.coefficients
    for b in range(B):
        q0 = np.concatenate([rng.uniform(-0.8, 0.8, size=3), [0.2 * b]])
        U = 3.0 * rng.standard_normal((N, 1))
        K = (0.2 * b + 0.9 * np.sin(2.0 * DT * np.arange(1, N + 1)))[:, None]
        r = rollout(system, q0, U, K, N, deriv_steps=(1, 50, N) if b == 0 else (), deriv2_full=True)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = q0
        arrays["b%d_U" % b] = U
        arrays["b%d_K" % b] = K
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], arrays["b0_U"], arrays["b0_K"], (0, 10, 100), seed=36)
    save("nonlinear_spring_arm", **arrays, **ds)


def gen_spring_link():
    """Synthetic system with LinearSpring potentials and a distance constraint.  The reference defines no third
    derivative for LinearSpring, so only first derivatives are recorded."""
    system = systems.spring_link(api=trep)
    rng = np.random.default_rng(20250 + 8)
    B, N = 2, 200
    arrays = dict(dt=DT, **topology(system))
    for b in range(B):
        q0 = np.concatenate([rng.uniform(-0.8, 0.8, size=3), [-1.0, 0.2 * b]])   # a, b, d, e, slide
        U = np.zeros((N, 0))
        K = (0.2 * b + 0.3 * np.sin(2.0 * DT * np.arange(1, N + 1)))[:, None]
        r = rollout(system, q0, U, K, N, deriv_steps=(1, 50, N) if b == 0 else (), second_order=False)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = q0
        arrays["b%d_K" % b] = K
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], np.zeros((N, 0)), arrays["b0_K"], (0, 10, 100), seed=17,
                          second_order=False)
    save("spring_link", **arrays, **ds)


def gen_plane_link():
    """Synthetic closed chain with PointOnPlane constraints on moving plane frames; starts at the zero configuration
    (both constraints satisfied) with seeded initial velocities made consistent by the first DEL step."""
    system = systems.plane_link(api=trep)
    B, N = 2, 200
    arrays = dict(dt=DT, **topology(system))
    U = np.zeros((N, 0)); K = np.zeros((N, 0))
    for b in range(B):
        q0 = np.zeros(system.nQ)
        r = rollout(system, q0, U, K, N, deriv_steps=(1, 50, N) if b == 0 else (), deriv2_full=True)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = q0
        system.get_potential(0) if False else None
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], U, K, (0, 10, 100), seed=18)
    save("plane_link", **arrays, **ds)


def gen_wrench_arm():
    """Synthetic system with HybridWrench forces: rollouts, first and full second derivative tensors, DSystem captures."""
    system = systems.wrench_arm(api=trep)
    rng = np.random.default_rng(20250 + 9)
    B, N = 2, 200
    arrays = dict(dt=DT, **topology(system))
    arrays["input_names"] = np.array([str(u.name) for u in system.inputs])
    for b in range(B):
        q0 = np.concatenate([rng.uniform(-0.8, 0.8, size=3), [0.2 * b]])
        U = rng.standard_normal((N, system.nu))
        K = (0.2 * b + 0.3 * np.sin(2.0 * DT * np.arange(1, N + 1)))[:, None]
        r = rollout(system, q0, U, K, N, deriv_steps=(1, 50, N) if b == 0 else (), deriv2_full=True)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = q0
        arrays["b%d_U" % b] = U
        arrays["b%d_K" % b] = K
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], arrays["b0_U"], arrays["b0_K"], (0, 10, 100), seed=19)
    save("wrench_arm", **arrays, **ds)


def gen_wrench_torque(name="wrench_torque", seed=11):
    """HybridWrench with torque components / SpatialWrench: rollouts, first and full second derivative tensors, DSystem
    captures."""
    system = getattr(systems, name)(api=trep)
    rng = np.random.default_rng(20250 + seed)
    B, N = 2, 200
    arrays = dict(dt=DT, **topology(system))
    arrays["input_names"] = np.array([str(u.name) for u in system.inputs])
    for b in range(B):
        q0 = np.concatenate([rng.uniform(-0.8, 0.8, size=3), [0.2 * b]])
        U = rng.standard_normal((N, system.nu))
        K = (0.2 * b + 0.3 * np.sin(2.0 * DT * np.arange(1, N + 1)))[:, None]
        r = rollout(system, q0, U, K, N, deriv_steps=(1, 50, N) if b == 0 else (), deriv2_full=True)
        for key, val in r.items():
            arrays["b%d_%s" % (b, key)] = val
        arrays["b%d_q0" % b] = q0
        arrays["b%d_U" % b] = U
        arrays["b%d_K" % b] = K
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], arrays["b0_U"], arrays["b0_K"], (0, 10, 100), seed=10 + seed)
    save(name, **arrays, **ds)


def gen_damper_link():
    """LinearDamper without LinearSpring: rollout, first and full second derivative tensors, DSystem captures."""
    system = systems.damper_link(api=trep)
    N = 200
    arrays = dict(dt=DT, **topology(system))
    q0 = system.q
    U = np.zeros((N, 0)); K = np.zeros((N, 0))
    r = rollout(system, q0, U, K, N, deriv_steps=(1, 50, N), deriv2_full=True)
    for key, val in r.items():
        arrays["b0_" + key] = val
    arrays["b0_q0"] = q0
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], U, K, (0, 10, 100), seed=23)
    save("damper_link", **arrays, **ds)


def gen_dual_pendulums():
    """examples/dual_pendulums.py run as the script does (tf = 10, dt = 0.01) plus first derivatives (the build has no
    second derivatives for LinearDamper)."""
    system = systems.dual_pendulums(api=trep)
    N = 1000
    arrays = dict(dt=DT, **topology(system))
    q0 = system.q
    U = np.zeros((N, 0)); K = np.zeros((N, 0))
    r = rollout(system, q0, U, K, N, deriv_steps=(1, 500), second_order=False)
    for key, val in r.items():
        arrays["b0_" + key] = val
    arrays["b0_q0"] = q0
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], U, K, (0, 10, 100), seed=22, second_order=False)
    save("dual_pendulums", **arrays, **ds)


def gen_puppet_forces():
    """Puppet(string_forces=True) under seeded random string forces: rollout and first derivatives."""
    system = systems.puppet_forces(api=trep)
    rng = np.random.default_rng(20250 + 10)
    N = 100
    arrays = dict(dt=DT, **topology(system))
    arrays["input_names"] = np.array([str(u.name) for u in system.inputs])
    system.q = 0.0
    system.q = systems.PUPPET_BASE_POSE
    q0 = system.q
    U = 0.5 * rng.standard_normal((N, system.nu))
    U[:, 2::3] += 3.0          # some lift on every hook
    K = np.zeros((N, 0))
    r = rollout(system, q0, U, K, N, deriv_steps=(1, 50), deriv2_select=("q2_dq1dq1", "p2_dq1du1", "q2_du1du1", "q2_dp1du1", "p2_dq1dq1"))
    for key, val in r.items():
        arrays["b0_" + key] = val
    arrays["b0_q0"] = q0
    arrays["b0_U"] = U
    ds = dsystem_captures(system, arrays["b0_Q"], arrays["b0_P"], U, K, (0, 10, 50), seed=20)
    save("puppet_forces", **arrays, **ds)


def gen_extensor_tendon():
    """examples/extensor-tendon-model.py run as the script does (tf = 10, dt = 0.01), plus first derivatives."""
    system = systems.extensor_tendon(api=trep)
    N = 1000
    arrays = dict(dt=DT, **topology(system))
    q0 = system.q
    U = np.zeros((N, 0)); K = np.zeros((N, 0))
    r = rollout(system, q0, U, K, N, deriv_steps=(1, 500), second_order=False)
    for key, val in r.items():
        arrays["b0_" + key] = val
    arrays["b0_q0"] = q0
    save("extensor_tendon", **arrays)


def gen_discopt_cart():
    """One DOptimizer trace on the pend-on-cart problem of examples/pend-on-cart-optimization.py:48-116
    (torque input enabled, 5 s horizon): a few quasi-Newton then Newton steps; per step the method,
    cost0, dcost0, accepted Armijo exponent, cost1 and the new trajectory."""
    from math import pi as mpi, cos
    system = systems.pend_on_cart(torque_force=True, api=trep)
    mvi = trep.MidpointVI(system, num_threads=1)
    t = np.arange(0.0, 5.0, DT)
    dsys = trep.discopt.DSystem(mvi, t)
    (X, U) = dsys.build_trajectory()
    for k in range(dsys.kf()):
        if k == 0:
            dsys.set(X[k], U[k], 0)
        else:
            dsys.step(U[k])
        X[k + 1] = dsys.f()
    amp = 130 * mpi / 180
    qd = np.zeros((len(t), system.nQ))
    th = system.get_config('theta').index
    for i, ti in enumerate(t):
        if 3.0 <= ti <= 7.0:
            qd[i, th] = (1 - cos(2 * mpi / 4 * (ti - 3.0))) * amp / 2
    (Xd, Ud) = dsys.build_trajectory(qd)
    wq = 0.01 * np.ones(dsys.nX); wq[system.get_config('x').index] = 0.01; wq[th] = 100.0
    wr = 0.01 * np.ones(dsys.nU)
    Qc, Rc = np.diag(wq), np.diag(wr)
    cost = trep.discopt.DCost(Xd, Ud, Qc, Rc)

    class Rec(trep.discopt.DOptimizerMonitor):
        def __init__(self):
            self.m = []
        def armijo_evaluation(self, armijo_iteration, nX, nU, bX, bU, cost, max_cost):
            self.m.append(armijo_iteration)
    mon = Rec()
    opt = trep.discopt.DOptimizer(dsys, cost, monitor=mon)
    out = dict(t=t, X0=X.copy(), U0=U.copy(), Xd=Xd, Ud=Ud, Q=Qc, R=Rc)
    methods = ['quasi', 'quasi', 'newton', 'newton', 'newton']
    (Kproj, dX, dU, Qm, Rm, Sm) = opt.calc_descent_direction(X, U, 'newton')
    out.update(dd_newton_dX=dX, dd_newton_dU=dU, dd_Kproj=np.array(Kproj),
               dd_newton_Q=np.array([Qm(k) for k in range(len(X))]),
               dd_newton_S=np.array([Sm(k) for k in range(len(X) - 1)]),
               dd_newton_R=np.array([Rm(k) for k in range(len(X) - 1)]))
    (Kproj, dX, dU, Qm, Rm, Sm) = opt.calc_descent_direction(X, U, 'quasi')
    out.update(dd_quasi_dX=dX, dd_quasi_dU=dU)
    out["cost_initial"] = np.array([opt.calc_cost(X, U)])
    for i, method in enumerate(methods):
        mon.m = []
        cost0 = opt.calc_cost(X, U)
        (done, X, U, dcost0, cost1) = opt.step(i, X, U, method)
        out["it%d_cost0" % i] = np.array([cost0]); out["it%d_dcost0" % i] = np.array([dcost0])
        out["it%d_cost1" % i] = np.array([cost1]); out["it%d_m" % i] = np.array([mon.m[-1] if mon.m else -1])
        out["it%d_X" % i] = X.copy(); out["it%d_U" % i] = U.copy()
    out["methods"] = np.array(methods)
    save("discopt_pend_on_cart", **out)


def gen_discopt_puppet(N=50):
    """One DOptimizer trace on the problem of examples/puppet-optimization.py (BASELINE config 4) at a short
    horizon: Puppet(string_constraints=True), desired trajectory = the four limb strings moving sinusoidally
    (:27-105), initial guess = strings held still (:127-155), weights QD 100, QK 1, PD 1, VK 1, RHO 0.1 (:20-24).
    Recorded: projection gain, Newton model Q/S/R at selected k, both descent directions, then one quasi-Newton and
    one Newton DOptimizer.step (cost0, dcost0, accepted Armijo exponent, cost1, new trajectory)."""
    system = systems.puppet(api=trep)
    nd, nk = system.nQd, system.nQk
    Q0 = systems.puppet_initial_conditions(system, 2, seed=20250 + 4)[1]
    t = DT * np.arange(N + 1)
    K_move = systems.puppet_string_schedule(system, Q0[None, nd:], N, DT)[0]
    K_still = np.repeat(Q0[None, nd:], N, axis=0)
    mvi = trep.MidpointVI(system, num_threads=1)
    dsys = trep.discopt.DSystem(mvi, t)
    x0 = dsys.build_state(Q0, np.zeros(nd), np.zeros(nk))

    def simulate(Uk):
        X = np.zeros((N + 1, dsys.nX))
        X[0] = x0
        for k in range(N):
            if k == 0:
                dsys.set(X[0], Uk[0], 0)
            else:
                dsys.step(Uk[k])
            X[k + 1] = dsys.f()
        return X
    Xd, Ud = simulate(K_move), K_move
    X, U = simulate(K_still), K_still.copy()
    wq = [100.0] * nd + [1.0] * nk + [1.0] * nd + [1.0] * nk
    Qc, Rc = np.diag(wq), np.diag([0.1] * nk)
    cost = trep.discopt.DCost(Xd, Ud, Qc, Rc)

    class Rec(trep.discopt.DOptimizerMonitor):
        def __init__(self):
            self.m = []
        def armijo_evaluation(self, armijo_iteration, nX, nU, bX, bU, cost, max_cost):
            self.m.append(armijo_iteration)
    mon = Rec()
    opt = trep.discopt.DOptimizer(dsys, cost, monitor=mon)
    ks = np.array([0, 1, N // 2, N - 1], dtype=np.int32)
    out = dict(t=t, q0=Q0, X0=X.copy(), U0=U.copy(), Xd=Xd, Ud=Ud, Q=Qc, R=Rc, model_k=ks)
    (Kproj, dX, dU, Qm, Rm, Sm) = opt.calc_descent_direction(X, U, 'newton')
    out.update(dd_newton_dX=dX, dd_newton_dU=dU, dd_Kproj=np.array(Kproj),
               dd_newton_Q=np.array([Qm(k) for k in ks]), dd_newton_Qf=Qm(N),
               dd_newton_S=np.array([Sm(k) for k in ks]), dd_newton_R=np.array([Rm(k) for k in ks]),
               dd_newton_dcost=np.array([opt.calc_dcost(X, U, dX, dU)]))
    (Kproj, dX, dU, Qm, Rm, Sm) = opt.calc_descent_direction(X, U, 'quasi')
    out.update(dd_quasi_dX=dX, dd_quasi_dU=dU, dd_quasi_dcost=np.array([opt.calc_dcost(X, U, dX, dU)]))
    out["cost_initial"] = np.array([opt.calc_cost(X, U)])
    methods = ['quasi', 'newton']
    for i, method in enumerate(methods):
        mon.m = []
        cost0 = opt.calc_cost(X, U)
        (done, X, U, dcost0, cost1) = opt.step(i, X, U, method)
        out["it%d_cost0" % i] = np.array([cost0]); out["it%d_dcost0" % i] = np.array([dcost0])
        out["it%d_cost1" % i] = np.array([cost1]); out["it%d_m" % i] = np.array([mon.m[-1] if mon.m else -1])
        out["it%d_X" % i] = X.copy(); out["it%d_U" % i] = U.copy()
        print("  puppet discopt it %d (%s): cost %.6f -> %.6f, dcost0 %.4e, m = %s" % (i, method, cost0, cost1, dcost0, out["it%d_m" % i]))
    out["methods"] = np.array(methods)
    save("discopt_puppet", **out)


def gen_discopt_cart_nonuniform():
    """The pend-on-cart problem of gen_discopt_cart on a NON-uniform time base (the reference's DSystem takes any time
    vector, dsystem.py:229-274): linearisation, projection gain, both descent directions and one quasi + one Newton step."""
    from math import pi as mpi, cos
    system = systems.pend_on_cart(torque_force=True, api=trep)
    mvi = trep.MidpointVI(system, num_threads=1)
    rng = np.random.default_rng(20250 + 77)
    dts = DT * (0.5 + rng.random(120))
    t = np.concatenate([[0.0], np.cumsum(dts)])
    dsys = trep.discopt.DSystem(mvi, t)
    (X, U) = dsys.build_trajectory()
    for k in range(dsys.kf()):
        if k == 0:
            dsys.set(X[k], U[k], 0)
        else:
            dsys.step(U[k])
        X[k + 1] = dsys.f()
    amp = 60 * mpi / 180
    qd = np.zeros((len(t), system.nQ))
    th = system.get_config('theta').index
    qd[:, th] = (1 - np.cos(2 * mpi * t / t[-1])) * amp / 2
    (Xd, Ud) = dsys.build_trajectory(qd)
    wq = 0.01 * np.ones(dsys.nX); wq[th] = 100.0
    Qc, Rc = np.diag(wq), np.diag(0.01 * np.ones(dsys.nU))
    cost = trep.discopt.DCost(Xd, Ud, Qc, Rc)

    class Rec(trep.discopt.DOptimizerMonitor):
        def __init__(self):
            self.m = []
        def armijo_evaluation(self, armijo_iteration, nX, nU, bX, bU, cost, max_cost):
            self.m.append(armijo_iteration)
    mon = Rec()
    opt = trep.discopt.DOptimizer(dsys, cost, monitor=mon)
    out = dict(t=t, X0=X.copy(), U0=U.copy(), Xd=Xd, Ud=Ud, Q=Qc, R=Rc)
    (A, B) = dsys.linearize_trajectory(X, U)
    out.update(lin_A=np.array(A), lin_B=np.array(B))
    (Kproj, dX, dU, Qm, Rm, Sm) = opt.calc_descent_direction(X, U, 'newton')
    out.update(dd_newton_dX=dX, dd_newton_dU=dU, dd_Kproj=np.array(Kproj))
    (Kproj, dX, dU, Qm, Rm, Sm) = opt.calc_descent_direction(X, U, 'quasi')
    out.update(dd_quasi_dX=dX, dd_quasi_dU=dU)
    methods = ['quasi', 'newton']
    for i, method in enumerate(methods):
        mon.m = []
        cost0 = opt.calc_cost(X, U)
        (done, X, U, dcost0, cost1) = opt.step(i, X, U, method)
        out["it%d_cost0" % i] = np.array([cost0]); out["it%d_dcost0" % i] = np.array([dcost0])
        out["it%d_cost1" % i] = np.array([cost1]); out["it%d_m" % i] = np.array([mon.m[-1] if mon.m else -1])
        out["it%d_X" % i] = X.copy(); out["it%d_U" % i] = U.copy()
    out["methods"] = np.array(methods)
    save("discopt_cart_nonuniform", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["known", "pend1", "pend5", "cart", "scissor", "puppet", "puppet_basic", "spring_arm", "nonlinear_spring_arm", "spring_link", "plane_link", "wrench_arm", "puppet_forces", "extensor_tendon", "wrench_torque", "dual_pendulums", "wrench_spatial", "wrench_body", "damper_link", "discopt", "discopt_puppet", "discopt_nonuniform"]
    if "known" in which:
        gen_known_answer()
    if "pend1" in which:
        gen_pendulum(1, 1000, "pendulum1")
    if "pend5" in which:
        gen_pendulum(5, 200, "pendulum5")
    if "cart" in which:
        gen_cart()
    if "scissor" in which:
        gen_scissor()
    if "puppet" in which:
        gen_puppet()
    if "puppet_basic" in which:
        gen_puppet_basic()
    if "puppet_forces" in which:
        gen_puppet_forces()
    if "extensor_tendon" in which:
        gen_extensor_tendon()
    if "damper_link" in which:
        gen_damper_link()
    if "dual_pendulums" in which:
        gen_dual_pendulums()
    if "wrench_torque" in which:
        gen_wrench_torque()
    if "wrench_spatial" in which:
        gen_wrench_torque("wrench_spatial", seed=12)
    if "wrench_body" in which:
        gen_wrench_torque("wrench_body", seed=13)
    if "wrench_arm" in which:
        gen_wrench_arm()
    if "plane_link" in which:
        gen_plane_link()
    if "spring_link" in which:
        gen_spring_link()
    if "spring_arm" in which:
        gen_spring_arm()
    if "nonlinear_spring_arm" in which:
        gen_nonlinear_spring_arm()
    if "discopt" in which:
        gen_discopt_cart()
    if "discopt_puppet" in which:
        gen_discopt_puppet()
    if "discopt_nonuniform" in which:
        gen_discopt_cart_nonuniform()
