#!/bin/bash
# SQ counter passes on the rollout kernel (diagnostic; run through gpurun): bash tools/collect_sq.sh <tag>
tag=${1:-sq}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $out/p$i.json 2> $out/p$i.err; echo "pass $i rc=$?"
  cp $(find $out/p$i -name "*counter_collection.csv" | head -1) $out/p$i.csv 2>/dev/null
  rm -rf $out/p$i
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections
for i in (1,2,3):
    try: rows=list(csv.DictReader(open("$out/p%d.csv"%i)))
    except Exception as e: print(i, e); continue
    acc=collections.defaultdict(list)
    for r in rows:
        if any(k in r['Kernel_Name'] for k in ('k_spec<0>', 'k_spec<0,', 'k_spec<(int)0')) or ('k_run' in r['Kernel_Name'] and ', 0>' in r['Kernel_Name']):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print("%-28s %16.0f  (n=%d)"%(k, sum(v)/len(v), len(v)))
PY
