#!/bin/bash
# A/B: current specialised kernel vs a hand-built alternative library (TREPAMD_SPEC_OVERRIDE=path)
for lib in "" "$PWD/trep_amd/_spec/old_gj.so"; do
  export TREPAMD_SPEC_OVERRIDE="$lib"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discopt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$lib]', d['value'], d['roofline']['kernel_avg_ms'], d['config']['kernel_variant'])"
done
