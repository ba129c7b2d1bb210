#!/usr/bin/env python3
"""numpy prototype of the WORLD-frame evaluation of the DEL residual's Lagrangian terms (round 5):

    L_ddq_k = s_k . H_k                      H_k = sum over the bodies below config k of their world-frame spatial momentum
    L_dq_k  = w_k . H_k + g . (M_k v_k + omega_k x C_k)

with s_k = world twist of joint k (velocity of the point at the world origin, angular velocity), V_k^- = sum of s_j dq_j over the
configs j above k on its path, w_k = [V_k^-, s_k], and M_k, C_k the mass and first moment of the subtree below k.  O(configs + bodies)
instead of the (body, config) items of system.c:129-202.  Checked here against the oracle's L_dq / L_ddq (reference system.c)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trep_amd import systems, descriptor          # noqa: E402
from oracle.oracle import OracleMVI               # noqa: E402


def bracket(a, b):      # [a, b] for twists (v, w)
    av, aw, bv, bw = a[:3], a[3:], b[:3], b[3:]
    return np.concatenate([np.cross(aw, bv) + np.cross(av, bw), np.cross(aw, bw)])


def world_eval(system, q, dq, grav):
    for c, x, v in zip(system.configs, q, dq):
        c.q, c.dq = x, v
    nq = len(system.configs)
    idx = {c: i for i, c in enumerate(system.configs)}
    s = np.zeros((nq, 6))
    path_of = {}
    for f in system.frames:
        c = f.config
        if c is None:
            continue
        g = f.g()
        kind = f.transform_type.name if hasattr(f.transform_type, "name") else str(f.transform_type)
        ax = {"x": 0, "y": 1, "z": 2}[kind[-1].lower()]
        a, p = g[:3, ax], g[:3, 3]
        if kind.lower().startswith("t"):
            s[idx[c]] = np.concatenate([a, np.zeros(3)])
        else:
            s[idx[c]] = np.concatenate([-np.cross(a, p), a])
        path_of[c] = [idx[x.config] for x in f._path() if x.config is not None]
    Vminus = np.zeros((nq, 6))
    for c, path in path_of.items():
        k = idx[c]
        for j in path[:-1]:
            Vminus[k] += s[j] * dq[j]
    w = np.array([bracket(Vminus[k], s[k]) for k in range(nq)])
    M = np.zeros(nq); C = np.zeros((nq, 3)); H = np.zeros((nq, 6))
    for f in system.masses:
        g = f.g()
        R, p = g[:3, :3], g[:3, 3]
        path = [idx[x.config] for x in f._path() if x.config is not None]
        V = sum(s[j] * dq[j] for j in path)
        m = f.mass
        I = R.dot(np.diag([f.Ixx, f.Iyy, f.Izz])).dot(R.T)
        lin = m * (V[:3] + np.cross(V[3:], p))
        ang = I.dot(V[3:]) + np.cross(p, lin)
        for j in path:
            M[j] += m; C[j] += m * p; H[j] += np.concatenate([lin, ang])
    L_ddq = np.einsum("kr,kr->k", s, H)
    G = M[:, None] * s[:, :3] + np.cross(s[:, 3:], C)
    L_dq = np.einsum("kr,kr->k", w, H) + G.dot(grav)
    return L_dq, L_ddq


def main():
    rng = np.random.default_rng(0)
    for name, make in (("puppet", systems.puppet), ("puppet_basic", systems.puppet_basic), ("scissor", lambda: systems.scissor_lift(4)),
                       ("pendulum5", lambda: systems.pendulum(5)), ("cart", systems.pend_on_cart)):
        system = make()
        d = descriptor.flatten(system)
        o = OracleMVI(d)
        nq = d.n_configs
        grav = None
        for pot in system.potentials:
            if hasattr(pot, "gravity"):
                grav = np.array(pot.gravity, dtype=float)
        worst = 0.0
        for _ in range(5):
            q = rng.uniform(-1, 1, nq); dq = rng.uniform(-2, 2, nq)
            if name.startswith("puppet"):
                q = systems.puppet_initial_conditions(system, 1, seed=int(rng.integers(1 << 30)))[0] if name == "puppet" else q
            Ldq, Lddq = world_eval(system, q, dq, grav)
            r = o.lagrangian(q, dq)
            scale = max(1.0, np.abs(r[0]).max(), np.abs(r[1]).max())
            worst = max(worst, np.abs(Ldq - r[0]).max() / scale, np.abs(Lddq - r[1]).max() / scale)
        print("%-14s nq=%2d  max |world form - oracle| (L_dq, L_ddq) = %.2e" % (name, nq, worst))


if __name__ == "__main__":
    main()
