#!/bin/bash
# round 5, first GPU call: baseline of the inherited code on this round's boxes + the wide parity sweep of the round-4 arithmetic
mkdir -p gpurun_out
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
echo "bench rc=$?"; head -c 600 gpurun_out/r05a_bench.json; echo
timeout 2400 python tools/stress_parity.py --seeds 3 --batch 128 --steps 200 > gpurun_out/r05a_stress.txt 2> gpurun_out/r05a_stress.err
echo "stress rc=$?"; tail -5 gpurun_out/r05a_stress.txt; tail -5 gpurun_out/r05a_stress.err
