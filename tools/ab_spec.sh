#!/bin/bash
# A/B of specialised rollout-kernel variants on ONE GPU box in ONE call: every argument is "name:extra compile flags" (name alone = the
# default flags); the variants are built here (hipcc, no GPU) into tools/ab/, then bench.py runs each of them REPS times, interleaved.
#   bash tools/ab_spec.sh base: sepC:-DTG_WEV_SEPARATE_C
cd "$(dirname "$0")/.."
mkdir -p tools/ab
REPS=${REPS:-2}
names=()
for v in "$@"; do
    name="${v%%:*}"; flags="${v#*:}"; [ "$flags" = "$v" ] && flags=""
    names+=("$name")
    TREPAMD_AB_NAME="$name" TREPAMD_AB_FLAGS="$flags" python - <<'PY' || exit 1
import os, shutil, sys
sys.path.insert(0, '.')
from trep_amd import specialize, systems
flags = specialize.DEFAULT_FLAGS + " " + os.environ["TREPAMD_AB_FLAGS"]
os.environ["TREPAMD_SPEC_FLAGS"] = flags
path = specialize.build(systems.puppet())
dst = os.path.join("tools", "ab", "lib_%s.so" % os.environ["TREPAMD_AB_NAME"])
shutil.copy(path, dst)
print(dst, "<-", flags)
PY
done
cmd="mkdir -p gpurun_out/ab; for r in \$(seq $REPS); do for n in ${names[*]}; do TREPAMD_SPEC_OVERRIDE=tools/ab/lib_\$n.so timeout 300 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-discopt > gpurun_out/ab/\$n.\$r.json 2> gpurun_out/ab/\$n.\$r.err; python -c \"import json; d=json.load(open('gpurun_out/ab/\$n.\$r.json')); print('%-12s rep %s  kernel %.2f ms  %.3f M steps/s  its/step %.3f  failed %d' % ('\$n', '\$r', d['roofline']['kernel_avg_ms'], d['value']/1e6, d['config']['newton_iterations_per_step'], d['config']['failed_trajectories']))\" || tail -3 gpurun_out/ab/\$n.\$r.err; done; done"
gpurun --timeout 1500 -- "$cmd" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)"
