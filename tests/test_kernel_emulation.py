"""CPU check of the DEVICE kernel source (trep_amd/csrc/mvi_core.hpp) compiled for the host with
TEAM = 1, against the reference golden vectors and the oracle.  The real parity tests (-m gpu) run
the same source on the MI355X through the C ABI; this file keeps the kernel logic covered where no
GPU exists."""
import numpy as np
import pytest

from common import BUILDERS, NO_SECOND_ORDER, build, golden, trajectories, relerr
from emu_harness import EmuBatch
from oracle.oracle import OracleMVI

DT = 0.01
TOL = 1e-10


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_emulated_rollout_matches_reference(name):
    g = golden(name)
    system, d = build(name)
    trajs = trajectories(name)
    B = len(trajs)
    n = len(g[trajs[0][0] + "IT"])
    e = EmuBatch(d, B)
    Q0 = np.array([t[1] for t in trajs])
    e.initialize_from_configs(0.0, Q0, DT, Q0)
    for b, (prefix, q0, U, K) in enumerate(trajs):
        assert relerr(e.p2[b], g[prefix + "P"][0]) < 1e-12
    U = np.array([t[2] for t in trajs])
    K = np.array([t[3] for t in trajs])
    X = e.rollout(n, DT, U, K)
    assert (e.status == 0).all()
    nq, nd = d.n_configs, d.n_dyn
    for b, (prefix, q0, _, _) in enumerate(trajs):
        assert relerr(X[b, :, :nq], g[prefix + "Q"]) < TOL, name
        assert relerr(X[b, :, nq:nq + nd], g[prefix + "P"]) < 1e-9, name
        assert relerr(e.lam[b], g[prefix + "LAM"][n]) < 1e-7
        assert abs(int(e.iters[b]) - int(g[prefix + "IT"].sum())) <= max(2, n // 100)


def test_emulated_residual_matches_oracle():
    system, d = build("puppet40")
    g = golden("puppet40")
    rng = np.random.default_rng(5)
    q1 = g["b0_Q"][10]
    q2 = q1.copy()
    q2[:d.n_dyn] += rng.uniform(-1e-3, 1e-3, d.n_dyn)
    q2[d.n_dyn:] = g["b0_K"][10]
    o = OracleMVI(d)
    o.q1, o.q2, o.p1 = q1, q2, g["b0_P"][10]
    o.lambda1 = g["b0_LAM"][10]
    o.set_times(0.1, 0.11)
    f_ref = o.calc_f()
    e = EmuBatch(d, 1)
    e.q1[0], e.q2[0], e.p1[0], e.lam[0] = q1, q2, g["b0_P"][10], g["b0_LAM"][10]
    e.t1, e.t2 = 0.1, 0.11
    f = e.calc_f()[0]
    assert relerr(f, f_ref) < 1e-12


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_emulated_first_derivatives_match_reference(name):
    from common import D1
    g = golden(name)
    system, d = build(name)
    prefix, q0, U, K = trajectories(name)[0]
    Q, P, LAM = g[prefix + "Q"], g[prefix + "P"], g[prefix + "LAM"]
    steps = sorted(int(k.split("_")[-3]) for k in g if k.startswith(prefix + "d1_") and k.endswith("q2_dq1"))
    for s_ in steps:
        k = s_ - 1
        e = EmuBatch(d, 1)
        e.t1 = e.t2 = (k + 1) * DT
        e.q1[0], e.q2[0], e.p1[0], e.p2[0], e.lam[0] = Q[k], Q[k], P[k], P[k], LAM[k]
        e.rollout(1, DT, U[None, k:k + 1], K[None, k:k + 1], want_X=False)
        assert relerr(e.q2[0], Q[k + 1]) < 1e-11
        out = e.deriv1()
        for n in D1:
            assert relerr(out[n][0], g["%sd1_%d_%s" % (prefix, s_, n)]) < 1e-9, (name, s_, n)
        # the same kernel writing DSystem.fdx / fdu directly (dsystem.py:284-317)
        A, B = e.linearize()
        nq, nd, nk, nu = d.n_configs, d.n_dyn, d.n_kin, d.n_inputs
        nX, nU, nqd = nq + nd + nk, nu + nk, nq + nd
        Ar, Br = np.zeros((nX, nX)), np.zeros((nX, nU))
        Ar[:nd, :nq], Ar[:nd, nq:nqd] = out["q2_dq1"][0].T, out["q2_dp1"][0].T
        Ar[nq:nqd, :nq], Ar[nq:nqd, nq:nqd] = out["p2_dq1"][0].T, out["p2_dp1"][0].T
        Ar[nqd:, nd:nq] = -np.eye(nk) / DT
        Br[:nd, :nu], Br[:nd, nu:] = out["q2_du1"][0].T, out["q2_dk2"][0].T
        Br[nq:nqd, :nu], Br[nq:nqd, nu:] = out["p2_du1"][0].T, out["p2_dk2"][0].T
        Br[nd:nq, nu:] = np.eye(nk)
        Br[nqd:, nu:] = np.eye(nk) / DT
        assert relerr(A[0], Ar) < 1e-12 and relerr(B[0], Br) < 1e-12, (name, s_)   # same numbers up to the summation order of the two output paths; 1/dt from t2 - t1


def oracle_hz(o, d, z):
    """z-contracted second derivatives assembled from the oracle's ten [A][B][out] tensor pairs."""
    nq, nd, nk, nu = d.n_configs, d.n_dyn, d.n_kin, d.n_inputs
    cnt = {"dq1": nq, "dp1": nd, "du1": nu, "dk2": nk}
    off = {"dq1": 0, "dp1": nq, "du1": nq + nd, "dk2": nq + nd + nu}
    R = nq + nd + nu + nk
    zq, zp = z[:nd], z[nq:nq + nd]
    HZ = np.zeros((R, R))
    names = ["dq1", "dp1", "du1", "dk2"]
    for ia, a in enumerate(names):
        for b in names[ia:]:
            if cnt[a] == 0 or cnt[b] == 0:
                continue
            blk = o.deriv2("q2_" + a + b) @ zq + o.deriv2("p2_" + a + b) @ zp
            HZ[off[a]:off[a] + cnt[a], off[b]:off[b] + cnt[b]] = blk
            HZ[off[b]:off[b] + cnt[b], off[a]:off[a] + cnt[a]] = blk.T
    return HZ


@pytest.mark.parametrize("name", sorted(set(BUILDERS) - NO_SECOND_ORDER))
def test_emulated_contracted_second_derivatives_match_oracle(name):
    g = golden(name)
    system, d = build(name)
    prefix, q0, U, K = trajectories(name)[0]
    Q, P, LAM = g[prefix + "Q"], g[prefix + "P"], g[prefix + "LAM"]
    rng = np.random.default_rng(3)
    k = 9
    e = EmuBatch(d, 1)
    e.t1 = e.t2 = (k + 1) * DT
    e.q1[0], e.q2[0], e.p1[0], e.p2[0], e.lam[0] = Q[k], Q[k], P[k], P[k], LAM[k]
    e.rollout(1, DT, U[None, k:k + 1], K[None, k:k + 1], want_X=False)
    o = OracleMVI(d)
    o.initialize_from_state((k + 1) * DT, Q[k], P[k], LAM[k])
    o.step((k + 2) * DT, U[k], K[k])
    o.calc_deriv2()
    z = rng.standard_normal(e.nX)
    HZ = e.deriv2z(z[None, :])[0]
    ref = oracle_hz(o, d, z)
    assert relerr(HZ, ref) < 1e-8, name
    if d.n_constraints:   # multiplier second derivatives: seed on the lambda rows (the reference's _l1_dAdB tensors)
        zl = rng.standard_normal(d.n_constraints)
        HL = e.deriv2z(np.zeros((1, e.nX)), zl[None, :])[0]
        nq, nd, nk, nu = d.n_configs, d.n_dyn, d.n_kin, d.n_inputs
        cnt = {"dq1": nq, "dp1": nd, "du1": nu, "dk2": nk}
        off = {"dq1": 0, "dp1": nq, "du1": nq + nd, "dk2": nq + nd + nu}
        names = ["dq1", "dp1", "du1", "dk2"]
        refl = np.zeros_like(HL)
        for ia, a in enumerate(names):
            for b in names[ia:]:
                if cnt[a] == 0 or cnt[b] == 0:
                    continue
                blk = o.deriv2("l1_" + a + b) @ zl
                refl[off[a]:off[a] + cnt[a], off[b]:off[b] + cnt[b]] = blk
                refl[off[b]:off[b] + cnt[b], off[a]:off[a] + cnt[a]] = blk.T
        assert relerr(HL, refl) < 1e-7, (name, relerr(HL, refl))


@pytest.mark.parametrize("links", [12, 34])
def test_emulated_long_chain_matches_oracle(links):
    """n-link pendulum (one chain of n joints, n(n+1)/2 items; 34 links exceed the 32-row register solver's range)."""
    from oracle.oracle import OracleMVI
    from trep_amd import systems, descriptor
    system = systems.pendulum(links)
    d = descriptor.flatten(system)
    rng = np.random.default_rng(links)
    q0 = rng.uniform(-0.6, 0.6, links)
    N = 10
    e = EmuBatch(d, 1)
    e.initialize_from_configs(0.0, q0[None], DT, q0[None])
    X = e.rollout(N, DT, np.zeros((1, N, 0)), np.zeros((1, N, 0)))
    o = OracleMVI(d)
    o.initialize_from_configs(0.0, q0, DT, q0)
    Xo, _ = o.rollout(N, DT, np.zeros((N, 0)), np.zeros((N, 0)))
    assert relerr(X[0], Xo) < 1e-10


def _star(n_arms, links_per_arm):
    """n_arms independent pendulum chains hanging off the world: many chains in one sweep round."""
    import trep_amd as T
    system = T.System()
    T.potentials.Gravity(system, name="Gravity")
    for a in range(n_arms):
        parent = system.world_frame
        for l in range(links_per_arm):
            joint = T.Frame(parent, T.RX if (a + l) % 2 == 0 else T.RY, "q%d_%d" % (a, l), "j%d_%d" % (a, l))
            parent = T.Frame(joint, T.TZ, -1.0 - 0.1 * a)
            parent.set_mass(1.0 + 0.05 * l, 0.1, 0.2, 0.3)
    return system


def test_emulated_star_matches_oracle():
    from oracle.oracle import OracleMVI
    from trep_amd import descriptor
    system = _star(20, 2)
    d = descriptor.flatten(system)
    rng = np.random.default_rng(8)
    q0 = rng.uniform(-0.5, 0.5, d.n_configs)
    N = 8
    e = EmuBatch(d, 1)
    e.initialize_from_configs(0.0, q0[None], DT, q0[None])
    X = e.rollout(N, DT, np.zeros((1, N, 0)), np.zeros((1, N, 0)))
    o = OracleMVI(d)
    o.initialize_from_configs(0.0, q0, DT, q0)
    Xo, _ = o.rollout(N, DT, np.zeros((N, 0)), np.zeros((N, 0)))
    assert relerr(X[0], Xo) < 1e-10


@pytest.mark.parametrize("name", ["pend_on_cart", "scissor4"])
def test_emulated_rollout_non_uniform_time_base(name):
    """RunArgs.dt_steps: one step size per step (the reference's DSystem takes any time vector, dsystem.py:229-274) --
    the kernel source under emulation against the oracle stepping through the same times."""
    from common import build, trajectories, relerr
    from emu_harness import EmuBatch
    from oracle.oracle import OracleMVI
    _, d = build(name)
    prefix, q0, U, K = trajectories(name)[0]
    N = 40
    rng = np.random.default_rng(3)
    dts = 0.01 * (0.5 + rng.random(N))
    e = EmuBatch(d, 1)
    e.initialize_from_configs(0.0, q0[None], 0.01, q0[None])
    X = e.rollout(N, dts[0], None if U.shape[1] == 0 else U[None, :N], None if K.shape[1] == 0 else K[None, :N], dts=dts)
    o = OracleMVI(d)
    o.initialize_from_configs(0.0, q0, 0.01, q0)
    nq = d.n_configs
    for k in range(N):
        o.step(o.times()[1] + dts[k], U[k], K[k])
        assert relerr(X[0, k + 1, :nq], o.q2) < 1e-10, k
    assert relerr(X[0, N, nq:nq + d.n_dyn], o.p2) < 1e-10
