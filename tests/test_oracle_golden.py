"""Pin the CPU oracle (oracle/trep_oracle.c) to golden vectors produced by the real reference."""
import numpy as np
import pytest

from common import BUILDERS, D1, NO_SECOND_ORDER, PAIRS, build, golden, trajectories, relerr
from oracle.oracle import OracleMVI, OracleError

TOL = 1e-10   # BASELINE.json north_star: fp64 state within 1e-10
DT = 0.01


def test_known_answer_single_pendulum():
    """The reference's own analytic check (examples/papers/tase2012/pend-single-step.py:32-41)."""
    import trep_amd as T
    g = golden("known_answer_pendulum")
    s = T.System()
    s.import_frames([T.rx('theta', name='pend_angle'), [T.tz(-1.0, name='pend_mass', mass=1.0)]])
    T.potentials.Gravity(s, (0, 0, -9.8))
    T.forces.ConfigForce(s, 'theta', 'theta-torque')
    o = OracleMVI(T.descriptor.flatten(s))
    o.initialize_from_state(0.0, [0.2], [0.5])
    assert o.step(0.1, [0.8]) == int(g["iterations"][0]) == 2
    assert abs(o.q2[0] - 0.24713619415556165) < 1e-14
    assert abs(o.q2[0] - 0.2471361941555716) < 1e-13   # hand-derived DEL root quoted in SURVEY.md §8c
    assert abs(o.p2[0] - g["p2"][0]) < 1e-14
    o.calc_deriv2()
    for n in ("q2_dq1", "q2_dp1", "q2_du1", "p2_dq1", "p2_dp1", "p2_du1"):
        assert relerr(o.deriv1(n), g[n]) < 1e-13, n
    for n in ("q2_dq1dq1", "p2_dq1dq1", "q2_du1du1", "q2_dq1du1", "p2_dq1du1", "q2_dp1dp1"):
        assert relerr(o.deriv2(n), g[n]) < 1e-12, n


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_free_running_rollout(name):
    g = golden(name)
    system, d = build(name)
    for prefix, q0, U, K in trajectories(name):
        o = OracleMVI(d)
        o.initialize_from_configs(0.0, q0, DT, q0)
        assert relerr(o.p2, g[prefix + "P"][0]) < 1e-13
        n = len(g[prefix + "IT"])
        its = []
        for k in range(n):
            its.append(o.step(o.t2 + DT, U[k], K[k]))
            assert relerr(o.q2, g[prefix + "Q"][k + 1]) < TOL, (name, prefix, k)
        # Newton iteration counts: identical except where the residual lands within rounding of
        # the 1e-10 stopping tolerance (then one count may differ by one; states still agree).
        diff = np.abs(np.array(its) - g[prefix + "IT"])
        assert diff.max() <= 1 and (diff != 0).mean() <= 0.01, (name, prefix, np.nonzero(diff)[0])
        assert relerr(o.p2, g[prefix + "P"][n]) < TOL
        assert relerr(o.lambda1, g[prefix + "LAM"][n]) < 1e-8


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_teacher_forced_steps(name):
    g = golden(name)
    system, d = build(name)
    prefix, q0, U, K = trajectories(name)[0]
    Q, P, LAM = g[prefix + "Q"], g[prefix + "P"], g[prefix + "LAM"]
    o = OracleMVI(d)
    for k in [0, 1, 2, 9, 49, len(Q) - 2]:
        o.initialize_from_state(k * DT + DT, Q[k], P[k], LAM[k])
        it = o.step((k + 2) * DT, U[k], K[k])
        assert abs(it - g[prefix + "IT"][k]) <= (0 if k < 100 else 1)
        assert relerr(o.q2, Q[k + 1]) < 1e-11
        assert relerr(o.p2, P[k + 1]) < 1e-11
        assert relerr(o.lambda1, LAM[k + 1]) < 1e-9


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_derivatives(name):
    g = golden(name)
    system, d = build(name)
    prefix, q0, U, K = trajectories(name)[0]
    Q, P, LAM = g[prefix + "Q"], g[prefix + "P"], g[prefix + "LAM"]
    steps = sorted(int(k.split("_")[-3]) for k in g if k.startswith(prefix + "d1_") and k.endswith("q2_dq1"))
    assert steps
    o = OracleMVI(d)
    for s in steps:
        k = s - 1
        o.initialize_from_state(k * DT + DT, Q[k], P[k], LAM[k])
        o.step((k + 2) * DT, U[k], K[k])
        if name in NO_SECOND_ORDER:
            with pytest.raises(OracleError):
                o.calc_deriv2()
            o.calc_deriv1()
        else:
            o.calc_deriv2()
        for n in D1:
            assert relerr(o.deriv1(n), g["%sd1_%d_%s" % (prefix, s, n)]) < 1e-10, (name, s, n)
        checked = 0
        for pr in PAIRS:
            for pre in ("q2_", "p2_", "l1_"):
                key = "%sd2_%d_%s%s" % (prefix, s, pre, pr)
                if key in g:
                    assert relerr(o.deriv2(pre + pr), g[key]) < 1e-9, (name, s, pre + pr)
                    checked += 1
        assert checked >= 5 or name in NO_SECOND_ORDER


def test_puppet_base_pose_first_step():
    """SURVEY.md Appendix A smoke values for the Puppet-40 base pose."""
    g = golden("puppet40")
    system, d = build("puppet40")
    o = OracleMVI(d)
    q0 = g["base_q0"]
    o.initialize_from_configs(0.0, q0, DT, q0)
    assert o.step(2 * DT, (), q0[d.n_dyn:]) == int(g["base_it"][0])
    assert relerr(o.q2, g["base_q2"]) < 1e-13
    assert relerr(o.p2, g["base_p2"]) < 1e-12
    assert relerr(o.lambda1, g["base_lambda1"]) < 1e-9


def test_residual_is_zero_after_solve():
    g = golden("scissor4")
    system, d = build("scissor4")
    prefix, q0, U, K = trajectories("scissor4")[0]
    o = OracleMVI(d)
    o.initialize_from_configs(0.0, q0, DT, q0)
    o.step(2 * DT, U[0], K[0])
    f = o.calc_f()
    assert np.linalg.norm(f[:d.n_dyn]) < 1e-10
    assert np.abs(f[d.n_dyn:]).max() < 1e-10
    assert relerr(f, g["b0_f_1"]) < 1e-9
