#!/usr/bin/env python3
"""examples/puppet-optimization.py of the reference (BASELINE config 4): the string-constrained marionette tracks a
desired motion (four limb strings moving sinusoidally, :27-105) starting from the trajectory with the strings held still
(:127-155), cost weights of :20-24, DOptimizer with the script's descent tolerance.  The trajectories go through the same
.mat files the script writes.  Horizon defaults to 2 s (the script uses 10 s; pass the number of seconds).

    python examples/puppet_optimization.py [seconds] [max_steps]
"""
import math
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd as trep
from trep_amd import discopt
from trep_amd.puppets import Puppet

tf = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dt = 0.01
QD_COST, QK_COST, PD_COST, VK_COST, RHO_COST = 100.0, 1.0, 1.0, 1.0, 0.1
workdir = tempfile.mkdtemp(prefix="puppet-opt-")


def generate_desired_trajectory(path):
    puppet = Puppet(joint_forces=False, string_forces=False, string_constraints=True)
    puppet.q = {'torso_rx': -0.05, 'torso_tz': 0.0, 'lelbow_rx': 1.57, 'relbow_rx': 1.57,
                'lhip_rx': math.pi / 2 - 0.6, 'rhip_rx': math.pi / 2 - 0.6,
                'lknee_rx': -math.pi / 2 + 0.6, 'rknee_rx': -math.pi / 2 + 0.6}
    puppet.project_string_controls()
    q0, qk0 = puppet.q, puppet.qk
    mvi = trep.MidpointVI(puppet)
    mvi.initialize_from_configs(0.0, q0, dt, q0)
    idx = dict((n, puppet.get_config(n + '_string-length').k_index) for n in ('left_leg', 'right_leg', 'left_arm', 'right_arm'))
    sign = {'left_leg': -1.0, 'right_leg': 1.0, 'left_arm': 1.0, 'right_arm': -1.0}
    nd = len(puppet.dyn_configs)
    q, p, v, rho, t = [mvi.q2], [mvi.p2], [np.zeros(puppet.nQk)], [], [mvi.t2]
    while mvi.t1 < tf:
        qk2 = np.array(qk0)
        for n in idx:
            qk2[idx[n]] += sign[n] * 0.1 * math.sin(0.6 * math.pi * mvi.t1)
        rho.append(qk2)
        mvi.step(mvi.t2 + dt, (), qk2)
        q.append(mvi.q2); p.append(mvi.p2); t.append(mvi.t2)
        v.append((np.asarray(mvi.q2) - np.asarray(mvi.q1))[nd:] / (mvi.t2 - mvi.t1))
    trep.save_trajectory(path, puppet, np.array(t), np.array(q), np.array(p), np.array(v), None, np.array(rho))


desired = os.path.join(workdir, 'puppet-desired.mat')
generate_desired_trajectory(desired)
system = Puppet(joint_forces=False, string_forces=False, string_constraints=True)
(t, Qd, p, v, u, rho) = trep.load_trajectory(desired, system)
dsys = discopt.DSystem(trep.MidpointVI(system), t)
(Xd, Ud) = dsys.build_trajectory(Qd)

# initial trajectory: strings held where the first desired pose puts them (:127-155)
Q0, p0, v0 = dsys.split_state(Xd[0])
system.q = Q0
system.project_string_controls()
system.correct_string_lengths()
x0 = dsys.build_state(system.q, p0, v0)
qk = system.qk
X, U = [x0], []
for k in range(len(Xd) - 1):
    if k == 0:
        dsys.set(X[0], qk, 0)
    else:
        dsys.step(qk)
    X.append(dsys.f()); U.append(qk)
initial = os.path.join(workdir, 'puppet-initial.mat')
dsys.save_state_trajectory(initial, np.array(X), np.array(U))

nd, nk = len(system.dyn_configs), len(system.kin_configs)
Qcost = np.diag([QD_COST] * nd + [QK_COST] * nk + [PD_COST] * nd + [VK_COST] * nk)
Rcost = np.diag([RHO_COST] * nk)
cost = discopt.DCost(Xd, Ud, Qcost, Rcost)
X, U = dsys.load_state_trajectory(initial)
optimizer = discopt.DOptimizer(dsys, cost, monitor=discopt.DOptimizerVerboseMonitor())
optimizer.descent_tolerance = 1e-2
optimizer.first_method_iterations = 2
finished, X, U = optimizer.optimize(X, U, max_steps=max_steps)
dsys.save_state_trajectory(os.path.join(workdir, 'puppet-result.mat'), X, U)
print("finished: %s, cost %.6f, files in %s" % (finished, optimizer.calc_cost(X, U), workdir))
