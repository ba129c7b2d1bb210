F="-DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form"
for v in wev nowev; do
  if [ $v = nowev ]; then export TREPAMD_SPEC_FLAGS="$F -DTG_NO_WEV"; else unset TREPAMD_SPEC_FLAGS; fi
  for shape in "32000 1" "8192 1" "960 200" "8192 200" "8192 5"; do
    set -- $shape
    timeout 300 python bench.py --batch $1 --rollout-steps $2 --steps 5 --warmup 2 --no-cpu-baseline --no-discopt > /tmp/b.json 2>/tmp/b.err
    python -c "
import json; d=json.load(open('/tmp/b.json'))
print('$v B=$1 N=$2  kernel %.3f ms  %.2f M steps/s  its/step %.3f' % (d['roofline']['kernel_avg_ms'], d['value']/1e6, d['config']['newton_iterations_per_step']))" || tail -3 /tmp/b.err
  done
done
