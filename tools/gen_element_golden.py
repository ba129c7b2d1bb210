#!/usr/bin/env python3
"""Golden vectors for the per-element queries of potentials (V ... V_dqdqdq), forces (f ... f_dudu) and constraints
(h ... h_dqdqdqdq) from the REAL reference (potential.py:42-76, force.py:46-145, constraint.py:56-102) at a seeded
state of the synthetic feature systems.  Build container only.  Writes tests/golden/elements.npz (data only): per
system the state and, per element and query, the full tensor over all config / input arguments."""
import itertools
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/tmp/trep_ref")

import trep  # noqa: E402
import trep.puppets  # noqa: E402
from trep_amd import systems  # noqa: E402

NAMES = ["pend_on_cart", "scissor4", "spring_arm", "nonlinear_spring_arm", "spring_link", "plane_link", "wrench_arm", "wrench_torque",
         "wrench_spatial", "wrench_body", "damper_link", "extensor_tendon", "puppet_basic"]
BUILD = {"pend_on_cart": systems.pend_on_cart, "scissor4": lambda api: systems.scissor_lift(4, api=api), "puppet_basic": systems.puppet_basic}
POT = [("V", 0), ("V_dq", 1), ("V_dqdq", 2), ("V_dqdqdq", 3)]
CON = [("h", 0), ("h_dq", 1), ("h_dqdq", 2), ("h_dqdqdq", 3), ("h_dqdqdqdq", 4)]
FORCE = [("f", "q"), ("f_dq", "qq"), ("f_ddq", "qq"), ("f_du", "qu"), ("f_dqdq", "qqq"), ("f_ddqdq", "qqq"), ("f_ddqddq", "qqq"),
         ("f_dudq", "quq"), ("f_duddq", "quq"), ("f_dudu", "quu")]


def tensor(fn, sets):
    shape = tuple(len(s) for s in sets)
    out = np.zeros(shape)
    for idx in itertools.product(*[range(n) for n in shape]):
        out[idx] = fn(*[s[i] for s, i in zip(sets, idx)])
    return out


def main():
    out = {}
    for name in NAMES:
        build = BUILD.get(name, None)
        system = build(api=trep) if build else getattr(systems, name)(api=trep)
        rng = np.random.default_rng(77)
        big = system.nQ > 12
        system.q = np.array(system.q) + 0.4 * rng.standard_normal(system.nQ)
        system.dq = rng.standard_normal(system.nQ)
        system.u = rng.standard_normal(system.nu)
        out[name + "_q"], out[name + "_dq"], out[name + "_u"] = np.array(system.q), np.array(system.dq), np.array(system.u)
        C, U = list(system.configs), list(system.inputs)
        for i, pot in enumerate(system.potentials):
            for acc, n in POT:
                if big and n > 2:
                    continue
                try:
                    out["%s_pot%d_%s" % (name, i, acc)] = tensor(getattr(pot, acc), [C] * n)
                except Exception as e:       # LinearSpring: no third derivative
                    out["%s_pot%d_%s_raises" % (name, i, acc)] = np.array([1])
        for i, con in enumerate(system.constraints):
            for acc, n in CON:
                if (big and n > 2) or (n > 3 and system.nQ > 6):
                    continue
                out["%s_con%d_%s" % (name, i, acc)] = tensor(getattr(con, acc), [C] * n)
        for i, force in enumerate(system.forces):
            for acc, sig in FORCE:
                if big and len(sig) > 2:
                    continue
                out["%s_force%d_%s" % (name, i, acc)] = tensor(getattr(force, acc), [C if ch == "q" else U for ch in sig])
        print(name, "potentials", len(system.potentials), "forces", len(system.forces), "constraints", len(system.constraints))
    # MidpointVI.discrete_fm2 / set_midpoint (midpointvi.c:430-482, 2710-2728) on the wrench arm
    ref = systems.wrench_arm(api=trep)
    rng = np.random.default_rng(3)
    q1 = np.array(ref.q) + 0.2 * rng.standard_normal(ref.nQ)
    q2 = q1 + 0.01 * rng.standard_normal(ref.nQ)
    u1 = rng.standard_normal(ref.nu)
    mvi = trep.MidpointVI(ref)
    mvi.initialize_from_configs(0.0, q1, 0.01, q2)
    mvi.u1 = u1
    out["fm2_q1"], out["fm2_q2"], out["fm2_u1"], out["fm2_value"] = q1, q2, u1, np.array(mvi.discrete_fm2())
    mvi.set_midpoint()
    out["fm2_mid_q"], out["fm2_mid_dq"], out["fm2_mid_t"] = np.array(ref.q), np.array(ref.dq), np.array([ref.t])
    # TapeMeasure through five frames of the extensor-tendon arm (tapemeasure.py:14-70)
    ref = systems.extensor_tendon(api=trep)
    rng = np.random.default_rng(9)
    ref.q = np.array(ref.q) + 0.3 * rng.standard_normal(ref.nQ)
    ref.dq = rng.standard_normal(ref.nQ)
    names = [f.name for f in ref.frames if f.name][1::3][:5]
    tape = trep.TapeMeasure(ref, names)
    C = list(ref.configs)
    out["tape_q"], out["tape_dq"] = np.array(ref.q), np.array(ref.dq)
    out["tape_frames"] = np.array(names)
    for acc, n in (("length", 0), ("length_dq", 1), ("length_dqdq", 2), ("length_dqdqdq", 3), ("velocity", 0), ("velocity_dq", 1),
                   ("velocity_dqdq", 2), ("velocity_ddq", 1), ("velocity_ddqdq", 2)):
        out["tape_" + acc] = tensor(getattr(tape, acc), [C] * n)
    path = os.path.join(REPO, "tests", "golden", "elements.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
