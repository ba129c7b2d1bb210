#!/bin/bash
# Round-5 judged measurements on the GPU box (run through gpurun):  bash tools/collect_r05.sh
# gpurun_out/r05/: bench.json, kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same command), pmc_fetch.csv /
# pmc_write.csv (separate --pmc passes), fp64.json, sq passes, discopt kernel stats, derivative-kernel throughputs.
tag=r05
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 10 --warmup 2 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --steps 10 --warmup 2 --no-specialize --no-discopt --no-cpu-baseline > $out/bench_generic.json 2> $out/bench_generic.err; echo "bench generic rc=$?"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_under_trace.json 2> $out/trace.err; echo "trace rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_rollout -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-discopt > $out/bench_under_trace_rollout.json 2> $out/trace_rollout.err; echo "rollout trace rc=$?"
cp $(find $out/trace_rollout -name "*kernel_stats.csv" | head -1) $out/kernel_stats_rollout.csv 2>/dev/null; rm -rf $out/trace_rollout
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-discopt > $out/bench_under_pmc_fetch.json 2> $out/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-discopt > $out/bench_under_pmc_write.json 2> $out/pmc_write.err; echo "pmc write rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
cp $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $out/pmc_fetch.csv 2>/dev/null
cp $(find $out/pmc_write -name "*counter_collection.csv" | head -1) $out/pmc_write.csv 2>/dev/null
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
BENCH_ARGS="--no-discopt" bash tools/collect_fp64.sh $tag/fp64 > $out/fp64.log 2>&1
BENCH_ARGS="--no-discopt" bash tools/collect_sq.sh $tag/sq > $out/sq.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_discopt -- python3 $GRAFT_REPO_ROOT/bench_discopt.py --seeds 256 --horizon 1000 --quasi 1 --newton 1 > $out/bench_discopt_under_trace.json 2> $out/trace_discopt.err; echo "discopt trace rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $out/trace_discopt -name "*kernel_stats.csv" | head -1) $out/kernel_stats_discopt.csv 2>/dev/null
rm -rf $out/trace_discopt
for s in 32 64 128 256; do timeout 600 python bench_discopt.py --seeds $s --horizon 1000 --quasi 1 --newton 1 --stages > $out/discopt_$s.json 2> $out/discopt_$s.err; done
for s in 32 64 128; do TREPAMD_NEWTON_PIPELINE=0 timeout 600 python bench_discopt.py --seeds $s --horizon 1000 --quasi 1 --newton 1 --stages > $out/discopt_${s}_unpipelined.json 2> $out/discopt_${s}_unpipelined.err; done
timeout 600 python tools/bench_derivs.py --batch 65536 > $out/bench_derivs.json 2>&1
TREPAMD_NO_SPECIALIZE=1 timeout 600 python tools/bench_derivs.py --batch 65536 > $out/bench_derivs_generic.json 2>&1
for sys in cart scissor puppet-basic; do timeout 600 python bench.py --system $sys --batch $([ $sys = puppet-basic ] && echo 8192 || echo 4096) --steps 10 --warmup 2 > $out/bench_$sys.json 2> $out/bench_$sys.err; done
python tools/time_lq.py > $out/lq_mfma.json 2>/dev/null
tools/micro/build.sh > /dev/null 2>&1; timeout 120 ./tools/micro/bin/mfma_f64_rate > $out/mfma_f64_rate.txt 2>&1
# the Newton solver in isolation: gj_rows vs gj_panel at the rollout kernel's occupancy (8 waves per CU) and alone (2 per CU)
for w in 8 4 2; do timeout 120 ./tools/micro/bin/gj_bench 28 2000 $w; done > $out/gj_bench.txt 2>&1
# the structured solve (gj_bbd) against gj_panel on puppet-pattern systems, same harness
for w in 8 2; do timeout 120 ./tools/micro/bin/bbd_bench 2000 $w; done > $out/bbd_bench.txt 2>&1
# VALU instruction classes of the rollout kernel (one more PMC pass)
cd /tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $out/pmc_cls1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-discopt > /dev/null 2> $out/pmc_cls1.err
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_cls2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-discopt > /dev/null 2> $out/pmc_cls2.err
cp $(find $out/pmc_cls1 -name "*counter_collection.csv" | head -1) $out/pmc_cls1.csv 2>/dev/null; cp $(find $out/pmc_cls2 -name "*counter_collection.csv" | head -1) $out/pmc_cls2.csv 2>/dev/null
rm -rf $out/pmc_cls1 $out/pmc_cls2
cd $GRAFT_REPO_ROOT
TREPAMD_LIB=trep_amd/libtrepamd_prof.so timeout 300 python tools/phase_profile_lq.py > $out/phase_profile_lq.txt 2>&1
timeout 300 python tools/step_latency.py > $out/step_latency.json 2> $out/step_latency.err
# phase profile of the specialised rollout kernel (diagnostic -DTG_PROFILE builds made on the build box: tools/collect_r05.sh expects them)
F="-DSPEC_ARGS_IN_MEMORY -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form -DTG_PROFILE"
(echo "# system-specialised rollout kernel, world-frame evaluation (profiling build: -DTG_PROFILE adds s_memtime stamps)"
 echo "# stamp names of the world form: 'attach+jacobians' = E3 (end points, world twists, body entries), 'attach+constraints' = E4 (constraints, list sums, momenta), 'velocities' = E5 (group composites), 'residual' = E6, 'newton init' = phase C, 'newton pairs' = phase D"
 TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F" timeout 200 python tools/phase_profile.py;
 echo; echo "# the (body, config) item form of round 4 with the composite Newton matrix (-DTG_NO_WEV)"; TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F -DTG_NO_WEV" timeout 200 python tools/phase_profile.py;
 echo; echo "# ... and with the (body, item, item) pair loop (-DTG_NO_WEV -DTG_NO_CMP)"; TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_SPEC_FLAGS="$F -DTG_NO_WEV -DTG_NO_CMP" timeout 200 python tools/phase_profile.py;
 echo; echo "# generic rollout kernel"; TREPAMD_LIB=trep_amd/libtrepamd_prof.so TREPAMD_NO_SPECIALIZE=1 timeout 200 python tools/phase_profile.py) > $out/phase_profile.txt 2>&1
bash tools/gpu_prof_derivs.sh $tag/prof_derivs > /dev/null 2>&1
ls -la $out; cat $out/bench.json | cut -c1-600; head -12 $out/kernel_stats.csv
