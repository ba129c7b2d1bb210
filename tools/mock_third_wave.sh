#!/bin/bash
# GPU-box half of tools/mock_third_wave.py: the five libraries on one box, twice each, interleaved
mkdir -p gpurun_out/r05
for rep in 1 2; do for n in real mock mock_w3 mock_lds mock_lds_w3; do
  TREPAMD_SPEC_OVERRIDE=tools/ab/lib_$n.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt > gpurun_out/r05/mock_$n.$rep.json 2> gpurun_out/r05/mock_$n.$rep.err
  python -c "
import json; d=json.load(open('gpurun_out/r05/mock_$n.$rep.json'))
print('%-12s rep $rep  kernel %.2f ms  %.2f M DEL-steps/s  its/step %.3f  lds/trajectory (host view) %d' % ('$n', d['roofline']['kernel_avg_ms'], d['value']/1e6, d['config']['newton_iterations_per_step'], d['config']['lds_bytes_per_trajectory']))" || tail -3 gpurun_out/r05/mock_$n.$rep.err
done; done
