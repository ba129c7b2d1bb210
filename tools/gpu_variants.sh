#!/bin/bash
# Compare experimental builds of the HIP library: rollout parity tests + one bench line each.
#   bash tools/gpu_variants.sh v0 v1 ...      (libtrepamd_<name>.so; "main" = the product library)
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = main ]; then unset TREPAMD_LIB; else export TREPAMD_LIB=$PWD/trep_amd/libtrepamd_$v.so; fi
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout_matches or random_batch or stepwise or full_size or long_chain or many_chains" > gpurun_out/var_${v}_pytest.log 2>&1
  echo "$v pytest rc=$? $(tail -1 gpurun_out/var_${v}_pytest.log)"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt > gpurun_out/var_${v}_bench.json 2> gpurun_out/var_${v}_bench.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/var_${v}_bench.json"))
    print("$v value %.4g steps/s  kernel %.2f ms  its/step %.3f  lds %s" % (d["value"], d["roofline"]["kernel_avg_ms"], d["config"]["newton_iterations_per_step"], d["config"]["lds_bytes_per_trajectory"]))
except Exception as e:
    print("$v no bench json", e); print(open("gpurun_out/var_${v}_bench.err").read()[-1500:])
PY
done
