#!/usr/bin/env python3
"""Build container only: replay, with the REFERENCE's DOptimizer, the optimiser steps that tools/dump_armijo_failures.py
saved on the GPU box (gpurun_out/armijo_failures.npz) -- the seeds of BASELINE config 4 (256 x N = 1000) whose Armijo
search the device-resident BatchDOptimizer flags as exhausted, plus one healthy control seed.  Question answered: does
the reference raise ConvergenceError("Armijo Failed to Converge") on exactly those iterates too?
Writes profiles/r02_armijo_reference_check.json.  TEST INFRASTRUCTURE (imports /tmp/trep_ref)."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/tmp/trep_ref")
import trep  # noqa: E402
import trep.discopt  # noqa: E402
import trep.puppets  # noqa: E402
from trep_amd import systems  # noqa: E402

d = np.load(os.path.join(REPO, "gpurun_out", "armijo_failures.npz"))
system = systems.puppet(api=trep)
mvi = trep.MidpointVI(system, num_threads=1)
dsys = trep.discopt.DSystem(mvi, d["t"])


class Rec(trep.discopt.DOptimizerMonitor):
    def __init__(self):
        self.m, self.sim_fail, self.info = [], [], []
    def armijo_evaluation(self, armijo_iteration, nX, nU, bX, bU, cost, max_cost):
        self.m.append((int(armijo_iteration), float(cost), float(max_cost)))
    def armijo_simulation_failure(self, armijo_iteration, nX, nU, bX, bU):
        self.sim_fail.append(int(armijo_iteration))
    def step_info(self, method, cost, dcost, X, U, dX, dU, Kproj):
        self.info.append((str(method), float(cost), float(dcost)))


out = {}
for tag in ("c", "f0", "f1"):
    if tag + "_X" not in d:
        continue
    mon = Rec()
    cost = trep.discopt.DCost(d[tag + "_Xd"], d[tag + "_Ud"], d["Q"], d["R"])
    opt = trep.discopt.DOptimizer(dsys, cost, monitor=mon)
    t0 = time.time()
    rec = {"seed": int(d[tag + "_seed"][0]), "iteration": int(d[tag + "_iteration"][0]), "method": str(d[tag + "_method"][0]),
           "device_cost0": float(d[tag + "_cost0"][0]), "device_dcost0": float(d[tag + "_dcost0"][0]),
           "device_final_method": str(d[tag + "_final_method"][0])}
    try:
        (done, X, U, dcost0, cost1) = opt.step(rec["iteration"], d[tag + "_X"], d[tag + "_U"], rec["method"])
        rec.update(reference_raised=False, reference_dcost0=float(dcost0), reference_cost1=float(cost1),
                   reference_armijo=mon.m[-1][0] if mon.m else None)
    except trep.ConvergenceError as e:
        rec.update(reference_raised=True, reference_error=str(e))
    rec["reference_step_info"] = mon.info
    rec["reference_armijo_evaluations"] = len(mon.m)
    rec["reference_armijo_simulation_failures"] = mon.sim_fail
    rec["reference_last_evaluations"] = mon.m[-3:]
    if tag == "c":
        rec.update(device_armijo=int(d["c_armijo"][0]), device_cost1=float(d["c_cost1"][0]))
    rec["seconds"] = time.time() - t0
    out[tag] = rec
    print(tag, json.dumps(rec))
    sys.stdout.flush()
with open(os.path.join(REPO, "profiles", "r02_armijo_reference_check.json"), "w") as fh:
    json.dump(out, fh, indent=1)
