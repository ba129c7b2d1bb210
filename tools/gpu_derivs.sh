#!/bin/bash
# derivative kernels: generic vs system-specialised
mkdir -p gpurun_out
for mode in generic spec; do
  if [ $mode = generic ]; then export TREPAMD_NO_SPECIALIZE=1; else unset TREPAMD_NO_SPECIALIZE; fi
  timeout 600 python tools/bench_derivs.py --batch 65536 > gpurun_out/derivs_$mode.json 2> gpurun_out/derivs_$mode.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/derivs_$mode.json"))
    print("$mode", {k: ("%.2f ms %.3g/s" % (v["kernel_ms"], v["per_s"])) for k, v in d.items() if isinstance(v, dict)})
except Exception as e:
    print("$mode failed", e); print(open("gpurun_out/derivs_$mode.err").read()[-1500:])
PY
done
timeout 900 python -m pytest tests -m gpu -x -q -k "deriv or second or lineariz or discopt or specialised" 2>&1 | tail -2
