#!/bin/bash
# LDS bank-conflict cycles per wave instruction of the rollout kernel's access patterns (tools/micro/lds_conflicts.hip), own --pmc pass.
# GPU box: bash tools/gpu_lds_conflicts.sh > gpurun_out/lds_conflicts.txt
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ldsc; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $out/p -- $GRAFT_REPO_ROOT/tools/micro/bin/lds_conflicts > $out/run.log 2> $out/run.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/p/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "rocclr" in r["Kernel_Name"]: continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["store b64, consecutive doubles (reference)", "store b64, item stride 6 doubles (J, W, X)", "store b64, joint / config / body-pose stride 12 doubles (local transforms, twists)", "store b64, image row stride 29",
         "store b128, stride 6", "store b128, stride 12", "load b64, consecutive (reference)", "load b64, stride 6", "load b64, stride 12", "load b64, stride 29 (image rows)",
         "load 2 x b64 (ds_read2_b64), stride 6", "load 2 x b64, stride 12", "load b128, stride 6 (what the compiler emits for an item's J / W)", "load b128, stride 12",
         "quad-lane sweep stores, round 0 pass 0 (puppet instances)", "quad-lane sweep stores, round 1 pass 0", "pair phase: twists of config a, b128 gather (12 a)", "pair phase: per-config vectors of b, b64 gather (15 b)", "pair phase: twists of a, b64 gather",
         "store b64, stride 13 doubles (12-double records padded by one: not 16-byte aligned)", "store b64, stride 14 doubles (padded by two: 16-byte aligned)", "store b128, stride 14",
         "load b64, stride 13", "load b64, stride 14", "load 2 x b64, stride 14", "load b128, stride 14"]
N = 2048.0 * 4096.0     # wave instructions per pattern (the volatile loads compile to flat loads: SQ_INSTS_LDS does not count them)
print("# LDS bank conflicts by access pattern (tools/micro/lds_conflicts.hip): 2048 waves x 4096 repetitions of ONE LDS instruction per pattern;")
print("# SQ_LDS_IDX_ACTIVE and SQ_LDS_BANK_CONFLICT per wave instruction (stores count 4 cycles, loads 2 per conflict-free 64 x 8 bytes)")
print("%-92s %10s %10s %8s" % ("pattern", "active", "conflict", "ratio"))
for i, d in enumerate(sorted(rows)):
    c = rows[d]
    print("%-92s %10.1f %10.1f %8.2f" % (names[i] if i < len(names) else str(d), c["SQ_LDS_IDX_ACTIVE"] / N, c["SQ_LDS_BANK_CONFLICT"] / N, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)))
PY
