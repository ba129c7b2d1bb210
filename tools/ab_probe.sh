F="-DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form"
for v in wev nowev; do
  if [ $v = nowev ]; then export TREPAMD_SPEC_FLAGS="$F -DTG_NO_WEV"; else unset TREPAMD_SPEC_FLAGS; fi
  echo "== $v"; timeout 600 python tools/probe_discopt_iters.py 2>&1 | tail -6
done
