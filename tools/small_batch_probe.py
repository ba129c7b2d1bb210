#!/usr/bin/env python3
"""Microseconds per batched DEL step of a 200-step device rollout at small batch sizes, and which rollout kernel ran.
    python tools/small_batch_probe.py [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd
from trep_amd import systems

DT = 0.01
pup = systems.puppet()
Q0 = systems.puppet_initial_conditions(pup, 1, seed=5)
Ks = systems.puppet_string_schedule(pup, Q0[:, pup.nQd:], 1200, DT)
for B in [int(a) for a in sys.argv[1:]] or [1, 64, 1024]:
    Qb = np.tile(Q0, (B, 1)); Kb = np.tile(Ks[:, :200], (B, 1, 1))
    mvi = trep_amd.BatchMidpointVI(pup, B)
    mvi.initialize_from_configs(0.0, Qb, DT, Qb)
    Kd = mvi.device_array(Kb)
    best = 1e9
    for rep in range(4):
        mvi.initialize_from_configs(0.0, Qb, DT, Qb)
        t0 = time.perf_counter()
        mvi.rollout_device(200, DT, None, Kd, None); mvi.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    it, st = mvi.status()
    print("B=%5d  %.1f us per batched step  its/step %.3f  ok %s  spec %s" % (B, best, it.mean() / 200, bool((st == 0).all()), os.path.basename(str(getattr(mvi, "_specialized", None)))))
