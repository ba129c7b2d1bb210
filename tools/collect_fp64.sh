#!/bin/bash
# fp64 VALU instruction counters of the rollout kernel (separate PMC pass; run through gpurun): bash tools/collect_fp64.sh <tag>
tag=${1:-fp64}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d $out/p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $out/bench.json 2> $out/err; echo "rc=$?"
cp $(find $out/p -name "*counter_collection.csv" | head -1) $out/fp64.csv 2>/dev/null
rm -rf $out/p
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections, json
rows=list(csv.DictReader(open("$out/fp64.csv")))
acc=collections.defaultdict(list)
for r in rows:
    if any(k in r['Kernel_Name'] for k in ('k_spec<0>', 'k_spec<0,', 'k_spec<(int)0')) or ('k_run' in r['Kernel_Name'] and ', 0>' in r['Kernel_Name']):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
res={k: sum(v)/len(v) for k,v in acc.items()}
print(json.dumps(res))
open("$out/fp64.json","w").write(json.dumps(res))
PY
tail -3 $out/err
