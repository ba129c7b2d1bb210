"""Shared helpers for the parity tests: golden fixtures + the systems they were made from."""
import os

import numpy as np

from trep_amd import systems, descriptor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

BUILDERS = {
    "pendulum1": lambda: systems.pendulum(1),
    "pendulum5": lambda: systems.pendulum(5),
    "pend_on_cart": lambda: systems.pend_on_cart(),
    "scissor4": lambda: systems.scissor_lift(4),
    "puppet40": lambda: systems.puppet(),
    "puppet_basic": lambda: systems.puppet_basic(),
    "spring_arm": lambda: systems.spring_arm(),
    "nonlinear_spring_arm": lambda: systems.nonlinear_spring_arm(),
    "spring_link": lambda: systems.spring_link(),
    "plane_link": lambda: systems.plane_link(),
    "wrench_arm": lambda: systems.wrench_arm(),
    "wrench_torque": lambda: systems.wrench_torque(),
    "dual_pendulums": lambda: systems.dual_pendulums(),
    "wrench_spatial": lambda: systems.wrench_spatial(),
    "wrench_body": lambda: systems.wrench_body(),
    "damper_link": lambda: systems.damper_link(),
    "puppet_forces": lambda: systems.puppet_forces(),
    "extensor_tendon": lambda: systems.extensor_tendon(),
}
D1 = ["q2_dq1", "q2_dp1", "q2_du1", "q2_dk2", "p2_dq1", "p2_dp1", "p2_du1", "p2_dk2",
      "l1_dq1", "l1_dp1", "l1_du1", "l1_dk2"]
PAIRS = ["dq1dq1", "dq1dp1", "dq1du1", "dq1dk2", "dp1dp1", "dp1du1", "dp1dk2", "du1du1", "du1dk2", "dk2dk2"]
NO_SECOND_ORDER = {"spring_link", "extensor_tendon", "dual_pendulums"}   # LinearSpring has no third derivative in the reference: deriv2 raises there too
_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return _cache[name]


def build(name):
    system = BUILDERS[name]()
    return system, descriptor.flatten(system)


def trajectories(name):
    """List of (prefix, q0, U[N][nu], K[N][nk]) recorded for a system."""
    g = golden(name)
    if name.startswith("pendulum"):
        return [("", g["q0"], g["U"], g["K"])]
    out = []
    b = 0
    while "b%d_q0" % b in g:
        n = len(g["b%d_IT" % b])
        U = g.get("b%d_U" % b, np.zeros((n, 0)))
        K = g.get("b%d_K" % b, np.zeros((n, 0)))
        out.append(("b%d_" % b, g["b%d_q0" % b], U, K))
        b += 1
    return out


def relerr(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
