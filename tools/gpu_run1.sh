#!/bin/bash
# Round-2 GPU call 1: full GPU suite, bench line with the discopt object, RCCL self-test, discopt seed sweep,
# and a dump of the seeds whose Armijo search fails at 256 x 1000 (checked against the reference in the container).
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r1_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r1_pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r1_bench.err
timeout 600 python bench.py --steps 3 --warmup 1 --force-dist --no-cpu-baseline --no-discopt > gpurun_out/r1_bench_dist.json 2> gpurun_out/r1_bench_dist.err; echo "bench dist rc=$?"; tail -c 600 gpurun_out/r1_bench_dist.err
for s in 32 64 128 256; do
  timeout 600 python bench_discopt.py --seeds $s --horizon 1000 --quasi 1 --newton 1 --stages > gpurun_out/r1_discopt_$s.json 2> gpurun_out/r1_discopt_$s.err; echo "discopt $s rc=$?"
done
timeout 900 python tools/dump_armijo_failures.py > gpurun_out/r1_armijo_dump.log 2>&1; echo "dump rc=$?"; tail -5 gpurun_out/r1_armijo_dump.log
python - <<'PY'
import json
for f in ["r1_bench", "r1_bench_dist"] + ["r1_discopt_%d" % s for s in (32, 64, 128, 256)]:
    try:
        d = json.load(open("gpurun_out/%s.json" % f))
        print(f, "value %.5g" % d["value"], d.get("roofline", {}).get("kernel_avg_ms"), (d.get("discopt") or {}).get("iters_per_s"), d.get("s_per_batched_newton_step"), d.get("armijo_failures"))
    except Exception as e:
        print(f, "no json", e)
PY
