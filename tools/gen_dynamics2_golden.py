#!/usr/bin/env python3
"""Golden vectors for the SECOND derivatives of the continuous dynamics from the REAL
reference: System.f_dqdq() ... f_dudu(), lambda_dqdq() ... lambda_dudu() (trep/system.py:982-1078, calc_dynamics_deriv2
system.c:1301-2029) at the states already recorded in tests/golden/dynamics.npz.  Build container only.
Writes tests/golden/dynamics2.npz (data only; the puppet's arrays are stored for one state)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/tmp/trep_ref")

import trep  # noqa: E402
import trep.puppets  # noqa: E402
from trep_amd import systems  # noqa: E402

BUILDERS = {
    "pendulum5": (lambda: systems.pendulum(5, api=trep), 2),
    "pend_on_cart": (lambda: systems.pend_on_cart(api=trep), 2),
    "scissor4": (lambda: systems.scissor_lift(4, api=trep), 2),
    "spring_arm": (lambda: systems.spring_arm(api=trep), 2),
    "plane_link": (lambda: systems.plane_link(api=trep), 2),
    "wrench_arm": (lambda: systems.wrench_arm(api=trep), 2),
    "wrench_torque": (lambda: systems.wrench_torque(api=trep), 1),
    "wrench_body": (lambda: systems.wrench_body(api=trep), 1),
    "damper_link": (lambda: systems.damper_link(api=trep), 1),
    "nonlinear_spring_arm": (lambda: systems.nonlinear_spring_arm(api=trep), 2),
    "puppet40": (lambda: systems.puppet(api=trep), 1),
}
NAMES = ["dqdq", "ddqdq", "ddqddq", "dddkdq", "dudq", "duddq", "dudu"]


def main():
    which = sys.argv[1:] or list(BUILDERS)
    g = dict(np.load(os.path.join(REPO, "tests", "golden", "dynamics.npz")))
    path = os.path.join(REPO, "tests", "golden", "dynamics2.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in which:
        build, n_states = BUILDERS[name]
        system = build()
        t0 = time.time()
        rec = dict(("%s_%s_%s" % (name, pre, n), []) for pre in ("f", "lam") for n in NAMES)
        for s in range(n_states):
            system.q, system.dq, system.u, system.ddqk = g[name + "_q"][s], g[name + "_dq"][s], g[name + "_u"][s], g[name + "_ddqk"][s]
            for n in NAMES:
                rec["%s_f_%s" % (name, n)].append(getattr(system, "f_" + n)())
                rec["%s_lam_%s" % (name, n)].append(getattr(system, "lambda_" + n)())
        out.update(dict((k, np.array(v)) for k, v in rec.items()))
        if name == "damper_link":
            # the reference's analytic f_ddqdq / f_dqdq of a LinearDamper disagree with its own first derivatives
            # (lineardamper.c:88); record central differences of the REFERENCE's f_ddq and f_dq as the consistent values
            q0, hh = g[name + "_q"][0].copy(), 1e-6
            fd_ddqdq = np.zeros_like(out[name + "_f_ddqdq"][0]); fd_dqdq = np.zeros_like(out[name + "_f_dqdq"][0])
            for j in range(len(q0)):
                vals = []
                for sgn in (+1, -1):
                    qq = q0.copy(); qq[j] += sgn * hh
                    system.q = qq
                    vals.append((system.f_ddq().copy(), system.f_dq().copy()))
                fd_ddqdq[:, j, :] = ((vals[0][0] - vals[1][0]) / (2 * hh)).T
                fd_dqdq[:, j, :] = ((vals[0][1] - vals[1][1]) / (2 * hh)).T
            out[name + "_fd_f_ddqdq"] = fd_ddqdq; out[name + "_fd_f_dqdq"] = fd_dqdq
        out[name + "_states"] = np.arange(n_states)
        print(name, "%.1f s" % (time.time() - t0), {k.split("_", 1)[1] if False else k: out[k].shape for k in rec if k.endswith("dqdq") and "_f_" in k})
        np.savez_compressed(path, **out)
    print("wrote %s (%.1f kB)" % (path, os.path.getsize(path) / 1e3))


def lagrangian_higher():
    """tests/golden/lagrangian_higher.npz: the reference's L_dqdqdq, L_ddqdqdq, L_ddqdqdqdq, L_ddqddqdq, L_ddqddqdqdq
    (system.py:869-949) for seeded index tuples (biased to the first 14 configs so that the puppet's tuples share kinematic chains)."""
    g = dict(np.load(os.path.join(REPO, "tests", "golden", "dynamics.npz")))
    out = {}
    rng = np.random.default_rng(5)     # systems are visited in a fixed order: appending one leaves the earlier tuples unchanged
    for name in ("pendulum5", "scissor4", "puppet40", "spring_arm", "plane_link", "nonlinear_spring_arm"):
        system = BUILDERS[name][0]()
        system.q, system.dq, system.u, system.ddqk = g[name + "_q"][0], g[name + "_dq"][0], g[name + "_u"][0], g[name + "_ddqk"][0]
        C = system.configs
        n = min(len(C), 14)
        idx = [tuple(int(x) for x in rng.integers(0, n, 4)) for _ in range(60)] + [(0, 0, 0, 0), (1, 1, 1, 1), (0, 1, 0, 1), (2, 1, 1, 2)]
        idx = [t for t in idx if max(t) < len(C)]
        vals = [[system.L_dqdqdq(C[a], C[b], C[c]), system.L_ddqdqdq(C[a], C[b], C[c]), system.L_ddqdqdqdq(C[a], C[b], C[c], C[d]),
                 system.L_ddqddqdq(C[a], C[b], C[c]), system.L_ddqddqdqdq(C[a], C[b], C[c], C[d])] for (a, b, c, d) in idx]
        out[name + "_idx"] = np.array(idx, dtype=np.int32)
        out[name + "_vals"] = np.array(vals)
        print(name, "largest |value| per accessor", np.abs(out[name + "_vals"]).max(axis=0), "non-zero", (np.abs(out[name + "_vals"]) > 1e-12).sum(axis=0))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "lagrangian_higher.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["lagrangian_higher"]:
        lagrangian_higher()
    else:
        main()
