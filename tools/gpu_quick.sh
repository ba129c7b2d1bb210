#!/bin/bash
# Quick GPU loop: rollout parity tests + one bench line (no CPU baseline).  Usage on the GPU box:
#   bash tools/gpu_quick.sh [tag]
tag=${1:-quick}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout_matches or random_batch or stepwise" > gpurun_out/${tag}_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${tag}_bench.json"))
    print("value %.4g steps/s  kernel %.1f ms  its/step %.3f  team %s lds %s" % (d["value"], d["roofline"]["kernel_avg_ms"], d["config"]["newton_iterations_per_step"], d["config"]["team"], d["config"]["lds_bytes_per_trajectory"]))
except Exception as e:
    print("no bench json", e); print(open("gpurun_out/${tag}_bench.err").read()[-2000:])
PY
