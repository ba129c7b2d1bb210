#!/usr/bin/env python3
"""examples/pendulum.py of the reference through the drop-in API (BASELINE config 1): an n-link pendulum built link by
link from rx / tz frames, integrated with MidpointVI for 10 s at dt = 0.01.

    python examples/pendulum.py [links]
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trep_amd as trep
from trep_amd import tx, ty, tz, rx, ry, rz  # noqa: F401  (the frame constructors of the reference's namespace)

links = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dt, tf, q0 = 0.01, 10.0, math.pi / 4.0


def make_pendulum(num_links):
    """Every link: a rotation about x driven by config link-i, then a unit mass one unit down the link (pendulum.py:36-52)."""
    def add_level(frame, link=0):
        if link == num_links:
            return
        child = trep.Frame(frame, trep.RX, "link-%d" % link, "link-%d" % link)
        end = trep.Frame(child, trep.TZ, -1.0, mass=1.0)
        add_level(end, link + 1)
    system = trep.System()
    trep.potentials.Gravity(system, (0, 0, -9.8))
    add_level(system.world_frame)
    return system


system = make_pendulum(links)
system.get_config("link-0").q = q0
mvi = trep.MidpointVI(system)
mvi.initialize_from_configs(0.0, system.q, dt, system.q)
q, t, iters = [mvi.q2], [mvi.t2], 0
while mvi.t1 < tf:
    iters += mvi.step(mvi.t2 + dt)
    q.append(mvi.q2)
    t.append(mvi.t2)
q = np.array(q)
print("%d-link pendulum: %d steps, %.2f Newton iterations per step, link-0 angle %.6f -> %.6f" %
      (links, len(t) - 1, iters / float(len(t) - 1), q[0, 0], q[-1, 0]))
