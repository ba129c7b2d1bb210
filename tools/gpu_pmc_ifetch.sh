#!/bin/bash
# Instruction-fetch counters of the rollout kernel (own --pmc pass, no trace domains).  GPU box: bash tools/gpu_pmc_ifetch.sh
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ifetch; mkdir -p $out
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-discopt > /dev/null 2> $out/p1.err
timeout 600 rocprofv3 --pmc SQ_IFETCH_LEVEL SQ_IFETCH SQC_TC_INST_REQ SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-discopt > /dev/null 2> $out/p2.err
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "k_spec" in r["Kernel_Name"] and "ILi0ELi0" in r["Kernel_Name"] or "k_spec<0, 0>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in sorted(acc): print(p, k, acc[k] / max(n[k], 1) , n[k])
PY
