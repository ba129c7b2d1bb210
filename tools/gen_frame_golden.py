#!/usr/bin/env python3
"""Golden vectors for the Frame kinematic accessors (frame.py:398-646: lg*, g*, g_inv*, p*, vb*) from the REAL
reference at a seeded state.  Build container only (imports the Python-3 working copy made by tools/build_reference.py).
Writes tests/golden/frames.npz (data only): for every system the state, and per accessor the stacked values over a fixed
enumeration of (frame, config tuple) cases that tests/test_frames.py reproduces."""
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

BUILDERS = {
    "pend_on_cart": lambda api: systems.pend_on_cart(api=api),
    "scissor4": lambda api: systems.scissor_lift(4, api=api),
    "spring_arm": lambda api: systems.spring_arm(api=api),
    "puppet40": lambda api: systems.puppet(api=api),
}
ORDER = {"g_dq": 1, "g_dqdq": 2, "g_dqdqdq": 3, "g_dqdqdqdq": 4, "g_inv_dq": 1, "g_inv_dqdq": 2,
         "p_dq": 1, "p_dqdq": 2, "p_dqdqdq": 3, "p_dqdqdqdq": 4,
         "vb_dq": 1, "vb_dqdq": 2, "vb_dqdqdq": 3, "vb_ddq": 1, "vb_ddqdq": 2, "vb_ddqdqdq": 3, "vb_ddqdqdqdq": 4}
PLAIN = ["lg", "lg_dq", "lg_dqdq", "lg_dqdqdq", "lg_dqdqdqdq", "lg_inv", "lg_inv_dq", "lg_inv_dqdq", "lg_inv_dqdqdq",
         "lg_inv_dqdqdqdq", "twist_hat", "g", "g_inv", "p", "vb"]


def cases(system, name, rng_seed=0):
    """(frame index, config index tuple) cases of accessor `name`: for every frame all tuples up to order 2, a seeded
    sample of the higher orders; tuples include configs the frame does not depend on."""
    n = ORDER[name]
    rng = np.random.default_rng(rng_seed + n)
    nq = system.nQ
    frames = range(len(system.frames))
    out = []
    for fi in frames:
        if n <= 2 and nq <= 12:
            tuples = list(itertools.product(range(nq), repeat=n))
        else:
            tuples = [tuple(rng.integers(0, nq, n)) for _ in range(6)]
            drv = [q.index for q in system.configs if system.frames[fi].uses_config(q)]
            if drv:
                tuples += [tuple(rng.choice(drv, n)) for _ in range(6)]
                tuples += [(d,) * n for d in drv[-3:]]          # repeated derivatives with respect to one driving config
        out += [(fi,) + tuple(int(x) for x in t) for t in tuples]
    return np.array(out, dtype=np.int64).reshape(len(out), n + 1)


def main():
    out = {}
    for sname, build in BUILDERS.items():
        system = build(trep)
        rng = np.random.default_rng(11)
        system.q = 0.7 * rng.standard_normal(system.nQ)
        system.dq = rng.standard_normal(system.nQ)
        out[sname + "_q"], out[sname + "_dq"] = np.array(system.q), np.array(system.dq)
        for name in PLAIN:
            out["%s_%s" % (sname, name)] = np.array([getattr(f, name)() for f in system.frames])
        for name in ORDER:
            cs = cases(system, name)
            vals = [getattr(system.frames[c[0]], name)(*[system.configs[i] for i in c[1:]]) for c in cs]
            out["%s_%s_cases" % (sname, name)] = cs
            out["%s_%s" % (sname, name)] = np.array(vals)
        print(sname, "frames", len(system.frames), "values", sum(v.size for k, v in out.items() if k.startswith(sname)))
    path = os.path.join(REPO, "tests", "golden", "frames.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
