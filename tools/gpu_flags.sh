#!/bin/bash
# compiler-flag variants of the specialised kernel (built in the container first), the generic-library variant, derivative kernels,
# phase profile and the GPU test suite in one call
BASE="-DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_INLINE -mllvm -disable-machine-licm"
bash tools/gpu_spec2.sh "$BASE" "$BASE -mllvm -amdgpu-sched-strategy=max-ilp" "$BASE -mllvm -amdgpu-sched-strategy=max-memory-clause" "$BASE -mllvm -amdgpu-sched-strategy=iterative-ilp" "$BASE -mllvm -enable-post-misched=0" "$BASE -mllvm -amdgpu-schedule-metric-bias=0" "$BASE -mllvm -disable-lsr"
unset TREPAMD_SPEC_FLAGS
echo "--- generic kernel: product build vs -disable-machine-licm"
for lib in libtrepamd.so libtrepamd_nolicm.so; do
  TREPAMD_LIB=trep_amd/$lib timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt --no-specialize 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['kernel_avg_ms'], d['value'])"
done
echo "--- derivative kernels"
timeout 600 python tools/bench_derivs.py --batch 65536 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: ('%.2f ms %.3g/s' % (v['kernel_ms'], v['per_s'])) for k, v in d.items() if isinstance(v, dict)})"
echo "--- phase profile (generic kernel, profiling build)"
TREPAMD_NO_SPECIALIZE=1 TREPAMD_LIB=trep_amd/libtrepamd_prof.so timeout 600 python tools/phase_profile.py
echo "--- gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
