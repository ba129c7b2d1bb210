#!/bin/bash
# A/B of the rollout kernel with the panel solver (default) and without it (-DTG_NO_GJ_PANEL specialisation), plus solver tests.
out=gpurun_out/${1:-ab}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "newton_solver or rollout_matches or random_batch or stepwise or specialised or exact_pivot or full_size_properties_puppet" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $out/pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt > $out/bench_panel.json 2> $out/bench_panel.err &
pid=$!
sleep 20; for i in 1 2 3; do /opt/rocm/bin/rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -i "power\|sclk\|mclk\|perf" | head -8; sleep 1; done > $out/smi.txt
wait $pid
TREPAMD_SPEC_FLAGS="-DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -DTG_NO_GJ_PANEL" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt > $out/bench_rows.json 2> $out/bench_rows.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt --no-specialize > $out/bench_generic.json 2> $out/bench_generic.err
python - <<PY
import json
for n in ("panel", "rows", "generic"):
    try:
        d = json.load(open("$out/bench_%s.json" % n))
        print(n, "%.4g steps/s kernel %.2f ms its/step %.3f failed %d %s" % (d["value"], d["roofline"]["kernel_avg_ms"], d["config"]["newton_iterations_per_step"], d["config"]["failed_trajectories"], d["config"]["spec_library"]))
    except Exception as e:
        print(n, "failed", e); print(open("$out/bench_%s.err" % n).read()[-1500:])
PY
cat $out/smi.txt | head -12
