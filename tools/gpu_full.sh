#!/bin/bash
# full GPU suite + default bench line
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/full_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/full_pytest.log | tail -3
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; echo "bench rc=$?"; tail -c 400 gpurun_out/full_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/full_bench.json"))
print("value %.5g  kernel %.2f ms  %s  discopt %.1f it/s  cpu %.0f" % (d["value"], d["roofline"]["kernel_avg_ms"], d["config"]["kernel_variant"], d["discopt"]["iters_per_s"], d["cpu_baseline"]["value"]))
PY
