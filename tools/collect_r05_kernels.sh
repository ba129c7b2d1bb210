#!/bin/bash
# One fp64 counter pass, one LDS pass and one timing pass for the kernels OTHER than the rollout (round-4 verdict, item 4): the specialised
# derivative kernels k_spec<3,0> / k_spec<4,0> and the continuous-dynamics kernels (tools/bench_derivs.py), the LQ sweep and the tangent
# rollout (bench_discopt.py at 32 seeds).  Output: gpurun_out/r05/kern_*.csv -> tools/summarize_r05_kernels.py
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|F64" | head -40 > $out/kern_counter_list.txt
P1="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for wl in derivs discopt; do
  if [ $wl = derivs ]; then cmd="python3 $GRAFT_REPO_ROOT/tools/bench_derivs.py --batch 65536 --reps 2"; else cmd="python3 $GRAFT_REPO_ROOT/bench_discopt.py --seeds 32 --horizon 1000 --quasi 1 --newton 1"; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$wl -- $cmd > $out/kern_${wl}_trace.json 2> $out/kern_${wl}_trace.err; echo "$wl trace rc=$?"
  cp $(find $out/t_$wl -name "*kernel_stats.csv" | head -1) $out/kern_${wl}_stats.csv 2>/dev/null; rm -rf $out/t_$wl
  i=0
  for set in "$P1" "$P2"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $set --output-format csv -d $out/p_$wl$i -- $cmd > /dev/null 2> $out/kern_${wl}_p$i.err; echo "$wl pmc $i rc=$?"
    python3 - <<PY
import csv, collections, glob, json
f = glob.glob("$out/p_$wl$i/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, dispatches=max(len(v) for v in cs.values())) for k, cs in acc.items()}
json.dump(res, open("$out/kern_${wl}_p$i.json", "w"), indent=0)
print(len(res), "kernels")
PY
    rm -rf $out/p_$wl$i
  done
done
ls -la $out | grep kern_
