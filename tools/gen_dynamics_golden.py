#!/usr/bin/env python3
"""Golden vectors for the continuous dynamics from the REAL reference.

Build container only (needs /tmp/trep_ref from tools/build_reference.py).  For each BASELINE system and a few
seeded states (q, dq, u, ddq of the kinematic configs) records System.f(), System.lambda_() and the first
derivatives f_dq, f_ddq, f_dddk, f_du, lambda_dq, lambda_ddq, lambda_dddk, lambda_du of the reference.
Writes tests/golden/dynamics.npz (data only).
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/tmp/trep_ref")

import trep  # noqa: E402
import trep.puppets  # noqa: E402
from trep_amd import systems  # noqa: E402

BUILDERS = {
    "pendulum5": lambda: systems.pendulum(5, api=trep),
    "pend_on_cart": lambda: systems.pend_on_cart(api=trep),
    "scissor4": lambda: systems.scissor_lift(4, api=trep),
    "puppet40": lambda: systems.puppet(api=trep),
    "puppet_basic": lambda: systems.puppet_basic(api=trep),
    "spring_arm": lambda: systems.spring_arm(api=trep),
    "spring_link": lambda: systems.spring_link(api=trep),
    "plane_link": lambda: systems.plane_link(api=trep),
    "wrench_arm": lambda: systems.wrench_arm(api=trep),
    "wrench_torque": lambda: systems.wrench_torque(api=trep),
    "dual_pendulums": lambda: systems.dual_pendulums(api=trep),
    "wrench_spatial": lambda: systems.wrench_spatial(api=trep),
    "wrench_body": lambda: systems.wrench_body(api=trep),
    "damper_link": lambda: systems.damper_link(api=trep),
    "nonlinear_spring_arm": lambda: systems.nonlinear_spring_arm(api=trep),
}
N_STATES = 4


def main():
    out = {}
    g = {n: np.load(os.path.join(REPO, "tests", "golden", n + ".npz")) for n in BUILDERS}
    for si, (name, build) in enumerate(BUILDERS.items()):
        system = build()
        rng = np.random.default_rng(4100 + si)
        # configurations that satisfy the constraints: states of the recorded rollouts
        Q = g[name]["Q"] if "Q" in g[name] else g[name]["b0_Q"]
        picks = np.linspace(1, len(Q) - 1, N_STATES).astype(int)
        nq, nd, nk, nu, nc = system.nQ, system.nQd, system.nQk, system.nu, system.nc
        qs, dqs, us, ddks = [], [], [], []
        for key in ("L_dq", "L_ddq", "L_dqdq", "L_ddqdq", "L_ddqddq", "E", "L", "f", "lam", "f_dq", "f_ddq", "f_dddk", "f_du", "lam_dq", "lam_ddq", "lam_dddk", "lam_du"):
            out["%s_%s" % (name, key)] = []
        for k in picks:
            q = Q[k][:nq]
            dq = 0.5 * rng.standard_normal(nq)
            u = rng.standard_normal(nu)
            ddk = rng.standard_normal(nk)
            system.q, system.dq, system.u, system.ddqk = q, dq, u, ddk
            qs.append(q); dqs.append(dq); us.append(u); ddks.append(ddk)
            C = system.configs
            out[name + "_L_dq"].append([system.L_dq(a) for a in C])
            out[name + "_L_ddq"].append([system.L_ddq(a) for a in C])
            out[name + "_L_dqdq"].append([[system.L_dqdq(a, b) for b in C] for a in C])
            out[name + "_L_ddqdq"].append([[system.L_ddqdq(a, b) for b in C] for a in C])
            out[name + "_L_ddqddq"].append([[system.L_ddqddq(a, b) for b in C] for a in C])
            out[name + "_E"].append(system.total_energy())
            out[name + "_L"].append(system.L())
            out[name + "_f"].append(system.f())
            out[name + "_lam"].append(system.lambda_())
            out[name + "_f_dq"].append(system.f_dq())
            out[name + "_f_ddq"].append(system.f_ddq())
            out[name + "_f_dddk"].append(system.f_dddk())
            out[name + "_f_du"].append(system.f_du())
            out[name + "_lam_dq"].append(system.lambda_dq())
            out[name + "_lam_ddq"].append(system.lambda_ddq())
            out[name + "_lam_dddk"].append(system.lambda_dddk())
            out[name + "_lam_du"].append(system.lambda_du())
        out[name + "_q"], out[name + "_dq"], out[name + "_u"], out[name + "_ddqk"] = qs, dqs, us, ddks
    out = dict((k, np.array(v)) for k, v in out.items())
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "dynamics.npz"), **out)
    for k in sorted(out):
        print(k, out[k].shape)


if __name__ == "__main__":
    main()
