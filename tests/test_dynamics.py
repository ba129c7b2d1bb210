"""Continuous dynamics ddq = f(q, dq, u, ddq_k), lambda (system.c:749-893).
Golden data: tools/gen_dynamics_golden.py (System.f() / System.lambda_() of the real reference)."""
import os

import numpy as np
import pytest

from common import GOLDEN, build, relerr

NAMES = ["pendulum5", "pend_on_cart", "scissor4", "puppet40", "puppet_basic", "spring_arm", "spring_link", "plane_link", "wrench_arm", "wrench_torque", "dual_pendulums", "wrench_spatial", "wrench_body", "damper_link", "nonlinear_spring_arm"]


def golden():
    return dict(np.load(os.path.join(GOLDEN, "dynamics.npz")))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_dynamics_match_reference(name):
    from oracle.oracle import OracleMVI
    g = golden()
    _, d = build(name)
    o = OracleMVI(d)
    for s in range(len(g[name + "_q"])):
        f, lam = o.dynamics(g[name + "_q"][s], g[name + "_dq"][s], g[name + "_u"][s], g[name + "_ddqk"][s])
        assert relerr(f, g[name + "_f"][s]) < 1e-11, (name, s)
        assert relerr(lam, g[name + "_lam"][s]) < 1e-11, (name, s)


@pytest.mark.parametrize("name", NAMES)
def test_emulated_kernel_dynamics_match_reference(name):
    """The device code of MODE_DYNAMICS compiled for the host (one lane) against the reference's f / lambda."""
    from emu_harness import EmuBatch
    g = golden()
    _, d = build(name)
    n = len(g[name + "_q"])
    e = EmuBatch(d, n)
    ddq, lam, status = e.dynamics(g[name + "_q"], g[name + "_dq"], g[name + "_u"], g[name + "_ddqk"])
    assert (status == 0).all()
    assert relerr(ddq, g[name + "_f"]) < 1e-10, name
    assert relerr(lam, g[name + "_lam"]) < 1e-10, name


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_dynamics_match_reference_and_oracle(name):
    """tg_batch_dynamics through the C ABI: golden states of the reference, then 256 random states against the oracle."""
    import trep_amd
    from oracle.oracle import OracleMVI
    g = golden()
    system, d = build(name)
    n = len(g[name + "_q"])
    eng = trep_amd.BatchMidpointVI(system, n)
    ddq, lam, status = eng.dynamics(g[name + "_q"], g[name + "_dq"], g[name + "_u"], g[name + "_ddqk"])
    assert (status == 0).all()
    assert relerr(ddq, g[name + "_f"]) < 1e-10 and relerr(lam, g[name + "_lam"]) < 1e-10
    eng.close()
    # the drop-in single-state API
    system.q, system.dq, system.u, system.ddqk = g[name + "_q"][1], g[name + "_dq"][1], g[name + "_u"][1], g[name + "_ddqk"][1]
    assert relerr(system.f(), g[name + "_f"][1]) < 1e-10
    assert relerr(system.lambda_(), g[name + "_lam"][1]) < 1e-10
    assert np.array_equal(system.ddqd, system.f())
    cfg = system.dyn_configs[-1]
    assert system.f(cfg) == system.f()[cfg.index]
    # a bigger batch of random states (off the constraint manifold as well: f is defined everywhere)
    rng = np.random.default_rng(9)
    B = 256
    Q = g[name + "_q"][rng.integers(0, n, B)] + 0.05 * rng.standard_normal((B, system.nQ))
    dQ = rng.standard_normal((B, system.nQ))
    U = rng.standard_normal((B, system.nu))
    ddK = rng.standard_normal((B, system.nQk))
    eng = trep_amd.BatchMidpointVI(system, B)
    ddq, lam, status = eng.dynamics(Q, dQ, U, ddK)
    o = OracleMVI(d)
    for b in range(0, B, 16):
        f_o, lam_o = o.dynamics(Q[b], dQ[b], U[b], ddK[b])
        assert relerr(ddq[b], f_o) < 1e-9 and relerr(lam[b], lam_o) < 1e-9, (name, b)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_dynamics_first_derivatives_match_reference(name):
    """to_dynamics_deriv1 (restating calc_dynamics_deriv1, system.c:912-1299) against System.f_dq() ... lambda_du()."""
    from oracle.oracle import OracleMVI
    g = golden()
    _, d = build(name)
    o = OracleMVI(d)
    for s in range(len(g[name + "_q"])):
        got = o.dynamics_deriv1(g[name + "_q"][s], g[name + "_dq"][s], g[name + "_u"][s], g[name + "_ddqk"][s])
        for key in ("f_dq", "f_ddq", "f_dddk", "f_du", "lam_dq", "lam_ddq", "lam_dddk", "lam_du"):
            ref = g["%s_%s" % (name, key)][s]
            assert got[key].shape == ref.shape, (name, key, got[key].shape, ref.shape)
            assert relerr(got[key], ref) < 1e-9, (name, s, key)


@pytest.mark.parametrize("name", NAMES)
def test_emulated_kernel_dynamics_first_derivatives_match_reference(name):
    from emu_harness import EmuBatch
    g = golden()
    _, d = build(name)
    n = len(g[name + "_q"])
    e = EmuBatch(d, n)
    got, status = e.dynamics_deriv1(g[name + "_q"], g[name + "_dq"], g[name + "_u"], g[name + "_ddqk"])
    assert (status == 0).all()
    for key in ("f_dq", "f_ddq", "f_dddk", "f_du", "lam_dq", "lam_ddq", "lam_dddk", "lam_du"):
        ref = g["%s_%s" % (name, key)]
        assert got[key].shape == ref.shape, (key, got[key].shape, ref.shape)
        assert relerr(got[key], ref) < 1e-9, (name, key)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_dynamics_first_derivatives_match_reference(name):
    """tg_batch_dynamics_deriv1 through the C ABI and the System.f_dq() ... lambda_du() accessors."""
    import trep_amd
    g = golden()
    system, d = build(name)
    n = len(g[name + "_q"])
    eng = trep_amd.BatchMidpointVI(system, n)
    got, status = eng.dynamics_deriv1(g[name + "_q"], g[name + "_dq"], g[name + "_u"], g[name + "_ddqk"])
    assert (status == 0).all()
    for key in ("f_dq", "f_ddq", "f_dddk", "f_du", "lam_dq", "lam_ddq", "lam_dddk", "lam_du"):
        ref = g["%s_%s" % (name, key)]
        mine = got[key.replace("lam_", "lambda_")]
        assert mine.shape == ref.shape, (key, mine.shape, ref.shape)
        assert relerr(mine, ref) < 1e-9, (name, key)
    eng.close()
    s = 2
    system.q, system.dq, system.u, system.ddqk = g[name + "_q"][s], g[name + "_dq"][s], g[name + "_u"][s], g[name + "_ddqk"][s]
    assert relerr(system.f_dq(), g[name + "_f_dq"][s]) < 1e-9
    assert relerr(system.f_ddq(), g[name + "_f_ddq"][s]) < 1e-9
    assert relerr(system.lambda_dq(), g[name + "_lam_dq"][s]) < 1e-9
    q0, qn = system.dyn_configs[0], system.configs[-1]
    assert abs(system.f_dq(q0, qn) - g[name + "_f_dq"][s][q0.index, qn.index]) < 1e-9 * max(1.0, np.abs(g[name + "_f_dq"][s]).max())
    # finite-difference cross-check of one column through the batch API (the reference's own style of validation)
    eng = trep_amd.BatchMidpointVI(system, 2)
    h = 1e-6
    Qp = np.tile(g[name + "_q"][s], (2, 1)); Qp[0, 0] += h; Qp[1, 0] -= h
    ddq, lam, st = eng.dynamics(Qp, g[name + "_dq"][s], g[name + "_u"][s], g[name + "_ddqk"][s])
    fd = (ddq[0] - ddq[1]) / (2 * h)
    assert relerr(fd, g[name + "_f_dq"][s][:, 0]) < 1e-5


@pytest.mark.parametrize("name", NAMES)
def test_oracle_energies_match_reference(name):
    from oracle.oracle import OracleMVI
    g = golden()
    _, d = build(name)
    o = OracleMVI(d)
    for s in range(len(g[name + "_q"])):
        T, V = o.energy(g[name + "_q"][s], g[name + "_dq"][s])
        assert abs((T + V) - g[name + "_E"][s]) < 1e-11 * max(1.0, abs(g[name + "_E"][s])), (name, s)
        assert abs((T - V) - g[name + "_L"][s]) < 1e-11 * max(1.0, abs(g[name + "_L"][s])), (name, s)


@pytest.mark.parametrize("name", NAMES)
def test_emulated_kernel_energies_match_reference(name):
    from emu_harness import EmuBatch
    g = golden()
    _, d = build(name)
    e = EmuBatch(d, len(g[name + "_q"]))
    TV = e.energy(g[name + "_q"], g[name + "_dq"])
    assert relerr(TV[:, 0] + TV[:, 1], g[name + "_E"]) < 1e-11 and relerr(TV[:, 0] - TV[:, 1], g[name + "_L"]) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_energies_match_reference(name):
    import trep_amd
    g = golden()
    system, d = build(name)
    n = len(g[name + "_q"])
    eng = trep_amd.BatchMidpointVI(system, n)
    TV = eng.energy(g[name + "_q"], g[name + "_dq"])
    assert relerr(TV[:, 0] + TV[:, 1], g[name + "_E"]) < 1e-11 and relerr(TV[:, 0] - TV[:, 1], g[name + "_L"]) < 1e-11
    eng.close()
    system.q, system.dq = g[name + "_q"][3], g[name + "_dq"][3]
    assert abs(system.total_energy() - g[name + "_E"][3]) < 1e-10 * max(1.0, abs(g[name + "_E"][3]))
    assert abs(system.L() - g[name + "_L"][3]) < 1e-10 * max(1.0, abs(g[name + "_L"][3]))


@pytest.mark.gpu
def test_system_derivative_validators():
    """System.test_derivative_dq / _ddq (system.py:1080-1203) applied to the continuous dynamics, the way the reference's
    own scripts validate f_dq / f_ddq."""
    g = golden()
    system, d = build("spring_link")
    system.q, system.dq, system.ddqk = g["spring_link_q"][1], g["spring_link_dq"][1], g["spring_link_ddqk"][1]
    assert system.test_derivative_dq(system.f, lambda q: system.f_dq(None, q), delta=1e-6, tolerance=1e-5)
    assert system.test_derivative_ddq(system.f, lambda q: system.f_ddq(None, q), delta=1e-6, tolerance=1e-6)
    assert system.test_derivative_dq(system.lambda_, lambda q: system.lambda_dq(None, q), delta=1e-6, tolerance=1e-5)
    assert not system.test_derivative_dq(system.f, lambda q: 2.0 * system.f_dq(None, q), delta=1e-6, tolerance=1e-5)


LAGRANGIAN_KEYS = ("L_dq", "L_ddq", "L_dqdq", "L_ddqdq", "L_ddqddq")


@pytest.mark.parametrize("name", NAMES)
def test_oracle_lagrangian_derivatives_match_reference(name):
    from oracle.oracle import OracleMVI
    g = golden()
    _, d = build(name)
    o = OracleMVI(d)
    for s in range(len(g[name + "_q"])):
        got = o.lagrangian(g[name + "_q"][s], g[name + "_dq"][s])
        for key, val in zip(LAGRANGIAN_KEYS, got):
            assert relerr(val, g["%s_%s" % (name, key)][s]) < 1e-11, (name, s, key)


@pytest.mark.parametrize("name", NAMES)
def test_emulated_kernel_lagrangian_derivatives_match_reference(name):
    from emu_harness import EmuBatch
    g = golden()
    _, d = build(name)
    e = EmuBatch(d, len(g[name + "_q"]))
    o1, o2 = e.lagrangian(g[name + "_q"], g[name + "_dq"])
    got = (o1[:, 0], o1[:, 1], o2[:, 0], o2[:, 1], o2[:, 2])
    for key, val in zip(LAGRANGIAN_KEYS, got):
        assert relerr(val, g["%s_%s" % (name, key)]) < 1e-11, (name, key)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_lagrangian_derivatives_match_reference(name):
    import trep_amd
    g = golden()
    system, d = build(name)
    n = len(g[name + "_q"])
    eng = trep_amd.BatchMidpointVI(system, n)
    got = eng.lagrangian(g[name + "_q"], g[name + "_dq"])
    for key in LAGRANGIAN_KEYS:
        assert relerr(got[key], g["%s_%s" % (name, key)]) < 1e-11, (name, key)
    eng.close()
    system.q, system.dq = g[name + "_q"][2], g[name + "_dq"][2]
    a, b = system.configs[0], system.configs[-1]
    ref = g[name + "_L_ddqdq"][2]
    assert abs(system.L_ddqdq(a, b) - ref[a.index, b.index]) < 1e-10 * max(1.0, np.abs(ref).max())
    assert abs(system.L_dq(b) - g[name + "_L_dq"][2][b.index]) < 1e-10 * max(1.0, np.abs(g[name + "_L_dq"][2]).max())
    assert abs(system.L_ddqddq(a, a) - g[name + "_L_ddqddq"][2][a.index, a.index]) < 1e-10 * max(1.0, np.abs(g[name + "_L_ddqddq"][2]).max())


# ---- second derivatives of the continuous dynamics (system.py:982-1078; calc_dynamics_deriv2, system.c:1301-2029) ---------
D2_NAMES = ["pendulum5", "pend_on_cart", "scissor4", "spring_arm", "plane_link", "wrench_arm", "wrench_torque", "wrench_body",
            "damper_link", "nonlinear_spring_arm", "puppet40"]
D2_KEYS = ["dqdq", "ddqdq", "ddqddq", "dddkdq", "dudq", "duddq", "dudu"]


def golden2():
    return dict(np.load(os.path.join(GOLDEN, "dynamics2.npz")))


def _check_second(name, got, g2, s, tol):
    for key in D2_KEYS:
        for pre, gpre in (("f", "f"), ("lambda", "lam")):
            ref = g2["%s_%s_%s" % (name, gpre, key)][s]
            a = got["%s_%s" % (pre, key)]
            assert a.shape == ref.shape, (name, pre, key, a.shape, ref.shape)
            # the reference's arrays everywhere -- including f_ddqdq with a LinearDamper (lineardamper.c:88) and f_dqdq with a
            # NonlinearConfigSpring (nonlinear_config_spring.c:56-60), whose element conventions System._apply_reference_conventions
            # carries into the dynamics exactly as calc_dynamics_deriv2 does
            assert relerr(a, ref) < tol, (name, s, pre, key, relerr(a, ref))


@pytest.mark.parametrize("name", D2_NAMES)
def test_emulated_dynamics_second_derivatives_match_reference(name):
    """The fourteen second-derivative arrays -- the analytic first-derivative kernel run on dual numbers (exact derivatives, no
    step size), here through the host emulation of that kernel -- against the reference's f_dqdq() ... lambda_dudu().  1e-10
    relative to each array's largest entry (or to 1); observed 1e-16 ... 4e-14."""
    from emu_harness import EmuBatch
    from trep_amd.system import dynamics_deriv2_forward
    g, g2 = golden(), golden2()
    _, d = build(name)
    rename = {"lam_dq": "lambda_dq", "lam_ddq": "lambda_ddq", "lam_dddk": "lambda_dddk", "lam_du": "lambda_du"}

    def deriv1_forward(Q, dQ, U, ddK, seed):
        e = EmuBatch(d, len(Q))
        out, status = e.dynamics_deriv1(Q, dQ, U, ddK, seeds=(seed,))
        assert (status == 0).all()
        return dict((rename.get(k, k), v) for k, v in out.items())
    system, _ = build(name)
    for s in g2[name + "_states"][:1 if name == "puppet40" else None]:
        q, dq, u, ddqk = g[name + "_q"][s], g[name + "_dq"][s], g[name + "_u"][s], g[name + "_ddqk"][s]
        got = dynamics_deriv2_forward(deriv1_forward, q, dq, u, ddqk)
        # the reference's element conventions (LinearDamper, NonlinearConfigSpring), with the emulated Lagrangian kernel's mass matrix
        system.q, system.dq, system.u, system.ddqk = q, dq, u, ddqk
        system._apply_reference_conventions(got, mass_matrix=EmuBatch(d, 1).lagrangian(q[None], dq[None])[1][0, 2])   # L2 = (L_dqdq, L_ddqdq, L_ddqddq)
        _check_second(name, got, g2, s, 1e-10)


def test_a_direction_on_no_variable_gives_zero_derivatives():
    """seed -1 (no direction): every forward-mode output is exactly zero; a seed on u reaches f_du's derivative only through the
    wrench's input columns (here: f_dq depends on u, f_du does not)."""
    from emu_harness import EmuBatch
    g = golden()
    name = "wrench_arm"
    _, d = build(name)
    q, dq, u, ddqk = g[name + "_q"][0], g[name + "_dq"][0], g[name + "_u"][0], g[name + "_ddqk"][0]
    e = EmuBatch(d, 2)
    rep = lambda a: np.repeat(a[None], 2, axis=0)
    nq, nk = len(q), len(ddqk)
    out, status = e.dynamics_deriv1(rep(q), rep(dq), rep(u), rep(ddqk), seeds=(np.array([-1, 2 * nq + nk], dtype=np.int32),))
    assert (status == 0).all()
    assert all(np.all(v[0] == 0.0) for v in out.values())
    assert np.abs(out["f_dq"][1]).max() > 0.0 and np.all(out["f_du"][1] == 0.0)


@pytest.mark.parametrize("name", ["pendulum5", "scissor4", "puppet40", "spring_arm", "plane_link"])
def test_emulated_higher_order_lagrangian_derivatives_match_reference(name):
    """Third- and fourth-order derivatives of the Lagrangian (System_L_dqdqdq ... L_ddqddqdqdq, system.c:204-622): the Lagrangian kernel
    on dual numbers with one direction (third order) and two nested ones (fourth), through the host emulation, against the reference's
    table look-ups for seeded index tuples.  1e-12 relative to the accessor's largest value over the tuples (or 1)."""
    from emu_harness import EmuBatch
    g = golden()
    gh = dict(np.load(os.path.join(GOLDEN, "lagrangian_higher.npz")))
    _, d = build(name)
    q, dq = g[name + "_q"][0], g[name + "_dq"][0]
    idx, ref = gh[name + "_idx"], gh[name + "_vals"]
    scale = np.maximum(1.0, np.abs(ref).max(axis=0))
    n = min(len(idx), 32)
    e = EmuBatch(d, n)
    Q, dQ = np.repeat(q[None], n, axis=0), np.repeat(dq[None], n, axis=0)
    s3, s4 = idx[:n, 2].astype(np.int32), idx[:n, 3].astype(np.int32)
    _, t3 = e.lagrangian(Q, dQ, seeds=(s3,))
    _, t4 = e.lagrangian(Q, dQ, seeds=(s3, s4))
    for m, ((a, b, _, _), r) in enumerate(zip(idx[:n], ref[:n])):
        got = np.array([t3[m, 0, a, b], t3[m, 1, a, b], t4[m, 1, a, b], t3[m, 2, a, b], t4[m, 2, a, b]])
        assert (np.abs(got - r) / scale < 1e-12).all(), (name, m, got, r)


@pytest.mark.gpu
@pytest.mark.parametrize("name", D2_NAMES)
def test_gpu_dynamics_second_derivatives_match_reference(name):
    """System.f_dqdq() ... lambda_dudu() (one launch of the forward-mode first-derivative kernel, one input variable per trajectory)
    against the reference at 1e-10, plus the accessors' object-indexed forms."""
    g, g2 = golden(), golden2()
    system, d = build(name)
    for s in g2[name + "_states"]:
        system.q, system.dq, system.u, system.ddqk = g[name + "_q"][s], g[name + "_dq"][s], g[name + "_u"][s], g[name + "_ddqk"][s]
        got = system._dynamics_deriv2()
        _check_second(name, got, g2, s, 1e-10)
    qd, q1, q2 = system.dyn_configs[0], system.configs[0], system.configs[-1]
    ref = g2[name + "_f_dqdq"][g2[name + "_states"][-1]]
    assert abs(system.f_dqdq(qd, q1, q2) - ref[q1.index, q2.index, qd.index]) < 1e-10 * max(1.0, np.abs(ref).max())
    assert system.f_ddqdq().shape == ref.shape
    if system.nc:
        c = system.constraints[0]
        refl = g2[name + "_lam_ddqddq"][g2[name + "_states"][-1]]
        assert abs(system.lambda_ddqddq(c, q1, q2) - refl[q1.index, q2.index, c.index]) < 1e-10 * max(1.0, np.abs(refl).max())


@pytest.mark.gpu
def test_gpu_forward_mode_entry_points_check_their_directions():
    """tg_batch_dynamics_deriv1_forward / tg_batch_lagrangian_forward: a direction variable outside q | dq | ddq_k | u is refused with
    TG_ERR_INVALID and nothing is launched; -1 (no direction) gives exact zeros; the batched call agrees with the emulated kernel."""
    import trep_amd
    from emu_harness import EmuBatch
    g = golden()
    name = "scissor4"
    system, d = build(name)
    q, dq, u, ddqk = g[name + "_q"][0], g[name + "_dq"][0], g[name + "_u"][0], g[name + "_ddqk"][0]
    nq, nk, nu = len(q), len(ddqk), len(u)
    nvar = 2 * nq + nk + nu
    B = 4
    eng = trep_amd.BatchMidpointVI(system, B)
    try:
        seeds = np.array([-1, 0, nq + 1, nvar - 1], dtype=np.int32)
        out, status = eng.dynamics_deriv1(q, dq, u, ddqk, seeds=(seeds,))
        assert (status == 0).all()
        assert all(np.all(v[0] == 0.0) for v in out.values())
        e = EmuBatch(d, B)
        rep = lambda a: np.repeat(np.asarray(a, dtype=float)[None], B, axis=0)
        ref, _ = e.dynamics_deriv1(rep(q), rep(dq), rep(u), rep(ddqk), seeds=(seeds,))
        for k, v in out.items():
            r = ref[k.replace("lambda_", "lam_")]
            assert relerr(v, r) < 1e-12, (k, relerr(v, r))
        for bad in (nvar, -2):
            with pytest.raises(Exception):
                eng.dynamics_deriv1(q, dq, u, ddqk, seeds=(np.array([0, 0, bad, 0], dtype=np.int32),))
            with pytest.raises(Exception):
                eng.lagrangian(q, dq, seeds=(np.zeros(B, dtype=np.int32), np.array([0, bad, 0, 0], dtype=np.int32)))
        with pytest.raises(ValueError):
            eng.lagrangian(q, dq, seeds=())
        third = eng.lagrangian(q, dq, seeds=(np.full(B, -1, dtype=np.int32),))
        assert all(np.all(v == 0.0) for v in third.values())
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pendulum5", "scissor4", "puppet40", "spring_arm", "plane_link", "nonlinear_spring_arm"])
def test_gpu_higher_order_lagrangian_accessors_match_reference(name):
    """System.L_dqdqdq, L_ddqdqdq, L_ddqddqdq (third order) and L_ddqdqdqdq, L_ddqddqdqdq (fourth order), system.py:869-949:
    the Lagrangian kernel on dual numbers (one direction / two nested ones) against the reference's table look-ups, for seeded index
    tuples.  1e-12 relative to the accessor's largest value over the tuples (or 1)."""
    g = golden()
    gh = dict(np.load(os.path.join(GOLDEN, "lagrangian_higher.npz")))
    system, d = build(name)
    system.q, system.dq, system.u, system.ddqk = g[name + "_q"][0], g[name + "_dq"][0], g[name + "_u"][0], g[name + "_ddqk"][0]
    C = system.configs
    idx, ref = gh[name + "_idx"], gh[name + "_vals"]
    scale = np.maximum(1.0, np.abs(ref).max(axis=0))
    worst = np.zeros(5)
    pick = list(range(24)) + ([i for i in range(len(idx)) if len(set(idx[i][:3])) == 1] if name == "nonlinear_spring_arm" else [])
    for (a, b, c, e), r in zip(idx[pick], ref[pick]):
        got = np.array([system.L_dqdqdq(C[a], C[b], C[c]), system.L_ddqdqdq(C[a], C[b], C[c]), system.L_ddqdqdqdq(C[a], C[b], C[c], C[e]),
                        system.L_ddqddqdq(C[a], C[b], C[c]), system.L_ddqddqdqdq(C[a], C[b], C[c], C[e])])
        worst = np.maximum(worst, np.abs(got - r) / scale)
    assert (worst < 1e-12).all(), worst
