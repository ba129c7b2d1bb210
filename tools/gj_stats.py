import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TREPAMD_SPEC_FLAGS"] = "-DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_STATS"
import trep_amd
from trep_amd import systems
s = systems.puppet(); B, N = 512, 50
Q0 = systems.puppet_initial_conditions(s, B, seed=3); K = systems.puppet_string_schedule(s, Q0[:, s.nQd:], N, 0.01)
m = trep_amd.BatchMidpointVI(s, B, specialize=True)
m.initialize_from_configs(0.0, Q0, 0.01, Q0); m.rollout(N, 0.01, None, K)
L = ctypes.CDLL(m._specialized)
out = (ctypes.c_uint64 * 8)(); L.tg_spec_gj_stats(out)
print("solves %d fallbacks %d singular-clause %d tie-clause %d last mask %x" % (out[0], out[1], out[2], out[3], out[4]))
