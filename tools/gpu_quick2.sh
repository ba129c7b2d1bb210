#!/bin/bash
# quick check of a kernel change: puppet bench line (specialised), parity subset, optional phase profile
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.4g steps/s  kernel %.2f ms  its/step %.3f failed %d %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['config']['newton_iterations_per_step'], d['config']['failed_trajectories'], d['config']['kernel_variant']))"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
if [ -n "$PROF" ]; then TREPAMD_NO_SPECIALIZE=1 TREPAMD_LIB=trep_amd/libtrepamd_prof.so timeout 600 python tools/phase_profile.py; fi
