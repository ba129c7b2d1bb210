"""Per-element queries of potentials (V ... V_dqdqdq), forces (f ... f_dudu) and constraints (h ... h_dqdqdqdq) against
tensors recorded from the reference (tests/golden/elements.npz, tools/gen_element_golden.py; potential.py:42-76,
force.py:46-145, constraint.py:56-102), plus the validate_* finite-difference checks.  Host-side queries: no GPU involved."""
import itertools

import numpy as np
import pytest

from common import golden

NAMES = ["pend_on_cart", "scissor4", "spring_arm", "nonlinear_spring_arm", "spring_link", "plane_link", "wrench_arm", "wrench_torque",
         "wrench_spatial", "wrench_body", "damper_link", "extensor_tendon", "puppet_basic"]


def _system(name):
    from trep_amd import systems
    if name == "scissor4":
        return systems.scissor_lift(4)
    return getattr(systems, name)()


def _tensor(fn, sets):
    shape = tuple(len(s) for s in sets)
    out = np.zeros(shape)
    for idx in itertools.product(*[range(n) for n in shape]):
        out[idx] = fn(*[s[i] for s, i in zip(sets, idx)])
    return out


@pytest.mark.parametrize("name", NAMES)
def test_element_queries_match_reference(name):
    g = golden("elements")
    system = _system(name)
    system.q, system.dq, system.u = g[name + "_q"], g[name + "_dq"], g[name + "_u"]
    C, U = list(system.configs), list(system.inputs)
    groups = {"pot": system.potentials, "con": system.constraints, "force": system.forces}
    checked = 0
    for key in sorted(k for k in g if k.startswith(name + "_") and k[len(name) + 1:].split("_")[0][:3] in ("pot", "con", "for")):
        tag, acc = key[len(name) + 1:].split("_", 1)
        kind = "pot" if tag.startswith("pot") else ("con" if tag.startswith("con") else "force")
        el = groups[kind][int(tag[len(kind):])]
        if acc.endswith("_raises"):
            with pytest.raises(NotImplementedError):
                getattr(el, acc[:-7])(*([C[0]] * 3))
            continue
        want = g[key]
        if kind == "force":
            sig = {"f": "q", "f_dq": "qq", "f_ddq": "qq", "f_du": "qu", "f_dqdq": "qqq", "f_ddqdq": "qqq", "f_ddqddq": "qqq",
                   "f_dudq": "quq", "f_duddq": "quq", "f_dudu": "quu"}[acc]
            sets = [C if ch == "q" else U for ch in sig]
        else:
            sets = [C] * want.ndim
        got = _tensor(getattr(el, acc), sets)
        assert got.shape == want.shape, key
        if want.size == 0:
            continue
        assert np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max()), (key, np.abs(got - want).max())
        checked += 1
    assert checked >= 4


def test_validate_methods_agree_with_their_own_derivatives():
    from trep_amd import systems
    system = systems.plane_link()
    rng = np.random.default_rng(5)
    system.q = np.array(system.q) + 0.2 * rng.standard_normal(system.nQ)
    for pot in system.potentials:
        assert pot.validate_V_dq() and pot.validate_V_dqdq() and pot.validate_V_dqdqdq(tolerance=1e-5)
    for con in system.constraints:
        assert con.validate_h_dq() and con.validate_h_dqdq() and con.validate_h_dqdqdq(tolerance=1e-5)
    system = systems.wrench_arm()
    system.q = np.array(system.q) + 0.2 * rng.standard_normal(system.nQ)
    system.dq = rng.standard_normal(system.nQ)
    system.u = rng.standard_normal(system.nu)
    for force in system.forces:
        assert force.validate_f_dq(tolerance=1e-5)


@pytest.mark.gpu
def test_set_midpoint_and_discrete_fm2_match_reference():
    import trep_amd
    from trep_amd import systems
    g = golden("elements")
    system = systems.wrench_arm()
    mvi = trep_amd.MidpointVI(system)
    mvi.initialize_from_configs(0.0, g["fm2_q1"], 0.01, g["fm2_q2"])
    mvi.u1 = g["fm2_u1"]
    fm2 = mvi.discrete_fm2()
    assert np.abs(fm2 - g["fm2_value"]).max() < 1e-12 * max(1.0, np.abs(g["fm2_value"]).max())
    mvi.set_midpoint()
    assert np.abs(np.array(system.q) - g["fm2_mid_q"]).max() < 1e-14 and np.abs(np.array(system.dq) - g["fm2_mid_dq"]).max() < 1e-11
    assert abs(system.t - g["fm2_mid_t"][0]) < 1e-15


def test_tape_measure_matches_reference():
    import trep_amd
    from trep_amd import systems
    g = golden("elements")
    system = systems.extensor_tendon()
    system.q, system.dq = g["tape_q"], g["tape_dq"]
    tape = trep_amd.TapeMeasure(system, [str(n) for n in g["tape_frames"]])
    C = list(system.configs)
    for acc in ("length", "length_dq", "length_dqdq", "length_dqdqdq", "velocity", "velocity_dq", "velocity_dqdq", "velocity_ddq", "velocity_ddqdq"):
        want = g["tape_" + acc]
        got = _tensor(getattr(tape, acc), [C] * want.ndim)
        assert np.abs(got - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), acc
    assert tape.validate_length_dq() and tape.validate_length_dqdq() and tape.validate_velocity_dq() and tape.validate_velocity_ddq()
    assert tape.validate_velocity_ddqdq() and tape.validate_velocity_dqdq(tolerance=1e-5) and tape.validate_length_dqdqdq(tolerance=1e-5)
