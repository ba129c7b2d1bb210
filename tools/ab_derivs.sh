#!/bin/bash
# A/B of specialised kernel variants on the DERIVATIVE kernels (tools/bench_derivs.py), one GPU call: arguments "name:extra flags".
#   bash tools/ab_derivs.sh base: nogj:-DTG_MOCK_D1=1
cd "$(dirname "$0")/.."
mkdir -p tools/ab
names=()
for v in "$@"; do
    name="${v%%:*}"; flags="${v#*:}"; [ "$flags" = "$v" ] && flags=""
    names+=("$name")
    (TREPAMD_AB_NAME="$name" TREPAMD_AB_FLAGS="$flags" python - <<'PY' || exit 1
import os, shutil, sys
sys.path.insert(0, '.')
from trep_amd import specialize, systems
os.environ["TREPAMD_SPEC_FLAGS"] = specialize.DEFAULT_FLAGS + " " + os.environ["TREPAMD_AB_FLAGS"]
path = specialize.build(systems.puppet())
dst = os.path.join("tools", "ab", "lib_%s.so" % os.environ["TREPAMD_AB_NAME"])
shutil.copy(path, dst)
print(dst, "<-", os.environ["TREPAMD_AB_FLAGS"])
PY
    ) &
done
wait
cmd="for n in ${names[*]}; do TREPAMD_SPEC_OVERRIDE=tools/ab/lib_\$n.so timeout 300 python tools/bench_derivs.py --batch ${BATCH:-65536} 2>&1 | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('%-10s' % '\$n', ' '.join('%s %.2f ms' % (k, d[k]['kernel_ms']) for k in ('step','deriv1_AB','deriv2z','dynamics_deriv1')))\"; done"
gpurun --timeout 1500 -- "$cmd" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)"
