#!/usr/bin/env python3
"""Drop-in latency of the B = 1 path: microseconds per `MidpointVI.step()` in the loop every reference example runs
(`while mvi.t1 < tf: mvi.step(mvi.t2 + dt); q.append(mvi.q2)`, examples/pendulum.py:73-86, puppet-basic.py:109-118),
next to the reference's own per-step time and to the same system in a batch.
    python tools/step_latency.py > profiles/r03_step_latency.json"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd
from trep_amd import systems

DT = 0.01
REFERENCE_US = {"pendulum-1": 12.0, "pend-on-cart": 12.0, "scissor-4": 212.0, "puppet-40": 2900.0}   # BASELINE.md section 2 (one Xeon core)


def loop(system, q0, n, u=None, k_fn=None, read_state=True):
    mvi = trep_amd.MidpointVI(system)
    mvi.initialize_from_configs(0.0, q0, DT, q0)
    def run(m):
        for i in range(m):
            kw = {}
            if u is not None:
                kw["u1"] = u
            if k_fn is not None:
                kw["k2"] = k_fn(mvi.t2 + DT)
            mvi.step(mvi.t2 + DT, **kw)
            if read_state:
                q = mvi.q2; p = mvi.p2
    run(20)
    t0 = time.perf_counter()
    run(n)
    return (time.perf_counter() - t0) / n * 1e6


def batch_per_step(system, Q0, K, n):
    B = len(Q0)
    mvi = trep_amd.BatchMidpointVI(system, B)
    mvi.initialize_from_configs(0.0, Q0, DT, Q0)
    Kd = mvi.device_array(K) if K is not None else None
    mvi.rollout_device(n, DT, None, Kd, None); mvi.synchronize()
    mvi.initialize_from_configs(0.0, Q0, DT, Q0)
    t0 = time.perf_counter()
    mvi.rollout_device(n, DT, None, Kd, None); mvi.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out = {"unit": "microseconds per MidpointVI.step() (host wall clock, including reading q2 and p2 back)", "dt": DT, "systems": {}}
pend = systems.pendulum(1)
out["systems"]["pendulum-1"] = {"step_us": loop(pend, [0.6], 2000), "step_only_us": loop(pend, [0.6], 2000, read_state=False)}
cart = systems.pend_on_cart()
out["systems"]["pend-on-cart"] = {"step_us": loop(cart, [0.1, 0.5], 2000, u=[0.3])}
sc = systems.scissor_lift(4)
out["systems"]["scissor-4"] = {"step_us": loop(sc, systems.scissor_q(sc, 0.2), 1000)}
pup = systems.puppet()
Q0 = systems.puppet_initial_conditions(pup, 1, seed=5)
Ks = systems.puppet_string_schedule(pup, Q0[:, pup.nQd:], 1200, DT)
idx = [0]
def k_fn(t):
    idx[0] += 1
    return Ks[0, min(idx[0], Ks.shape[1] - 1)]
out["systems"]["puppet-40"] = {"step_us": loop(pup, Q0[0], 1000, k_fn=k_fn)}
for name, d in out["systems"].items():
    d["reference_trep_us"] = REFERENCE_US[name]
    d["speedup_at_B1"] = REFERENCE_US[name] / d["step_us"]
# crossover: batch size at which one device step of the whole batch costs what B reference steps cost
for name, system, q0, K in (("pendulum-1", pend, np.array([[0.6]]), None), ("puppet-40", pup, Q0, Ks)):
    rows = {}
    for B in (1, 4, 16, 64, 256, 1024):
        Qb = np.tile(q0, (B, 1)); Kb = None if K is None else np.tile(K[:, :200], (B, 1, 1))
        us = batch_per_step(system, Qb, Kb, 200)
        rows[B] = {"us_per_batched_step_in_a_200_step_rollout": us, "us_per_trajectory_step": us / B}
    out["systems"][name]["batched_rollout"] = rows
    out["systems"][name]["crossover_batch_vs_reference"] = next((B for B in rows if rows[B]["us_per_trajectory_step"] < REFERENCE_US[name]), None)
print(json.dumps(out, indent=1))
