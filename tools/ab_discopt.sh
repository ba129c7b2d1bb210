F="-DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form"
for v in wev nowev wev nowev; do
  for s in 32 256; do
    if [ $v = nowev ]; then export TREPAMD_SPEC_FLAGS="$F -DTG_NO_WEV"; else unset TREPAMD_SPEC_FLAGS; fi
    timeout 600 python bench_discopt.py --seeds $s --horizon 1000 --quasi 1 --newton 1 --stages > /tmp/d.json 2>/tmp/d.err
    python -c "
import json; d=json.load(open('/tmp/d.json')); st=d['stage_seconds']
print('$v', $s, '%.1f it/s' % d['iters_per_s'], {k[:14]: round(x,4) for k,x in st.items()})"
  done
done
