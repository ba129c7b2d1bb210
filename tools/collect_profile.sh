#!/bin/bash
# Collect the judged measurements on the GPU box (run through gpurun):
#   bash tools/collect_profile.sh r01
# Produces under gpurun_out/<tag>/: bench.json (with cpu_baseline), kernel_stats.csv (rocprofv3
# --kernel-trace --stats of the same bench command), pmc_fetch.csv / pmc_write.csv (separate PMC
# passes, as MI355X_MICROARCH.md §HBM prescribes), phase_profile.txt (diagnostic build).
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 5 --warmup 1 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_under_trace.json 2> $out/trace.err; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_under_pmc_fetch.json 2> $out/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_under_pmc_write.json 2> $out/pmc_write.err; echo "pmc write rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
cp $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $out/pmc_fetch.csv 2>/dev/null
cp $(find $out/pmc_write -name "*counter_collection.csv" | head -1) $out/pmc_write.csv 2>/dev/null
if [ -f trep_amd/libtrepamd_prof.so ]; then TREPAMD_LIB=trep_amd/libtrepamd_prof.so python tools/phase_profile.py 8192 50 > $out/phase_profile.txt 2>&1; fi
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
ls -la $out; cat $out/bench.json; cat $out/kernel_stats.csv; head -5 $out/pmc_fetch.csv; head -5 $out/pmc_write.csv
# kernel-level view of one batched discopt run (all kernels of the device-resident optimiser)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_discopt -- python3 $GRAFT_REPO_ROOT/bench_discopt.py --seeds 64 --horizon 400 --quasi 1 --newton 1 > $out/bench_discopt_under_trace.json 2> $out/trace_discopt.err; echo "discopt trace rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $out/trace_discopt -name "*kernel_stats.csv" | head -1) $out/kernel_stats_discopt.csv 2>/dev/null
rm -rf $out/trace_discopt
timeout 900 python bench_discopt.py --seeds 256 --horizon 1000 --quasi 2 --newton 2 --stages > $out/bench_discopt.json 2> $out/bench_discopt.err
python tools/bench_derivs.py --batch 65536 > $out/bench_derivs.json 2>&1
cat $out/kernel_stats_discopt.csv | head -20; cat $out/bench_discopt.json | cut -c1-300
