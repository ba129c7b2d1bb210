#!/bin/bash
# Register / scratch / spill figures of every k_run instantiation (hipcc -Rpass-analysis=kernel-resource-usage, flags of csrc/Makefile).
#   tools/resource_usage.sh ["extra flags"] [filter]
cd "$(dirname "$0")/../trep_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm $1 -Rpass-analysis=kernel-resource-usage -c -o /dev/null trepamd.hip 2>&1 | \
python3 -c '
import re, sys
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", line)
    if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else "k_runILi64"
for r in rows:
    if flt in r["name"]:
        n = re.sub(r"_ZN12_GLOBAL__N_15k_runILi(\d+)ELi(\d+)ELb(\d)E.*", r"k_run<\1,\2,\3>", r["name"])
        print("%-18s VGPR %3d AGPR %3d SGPR %3d  sgpr-spill %4d vgpr-spill %3d scratch %4d  occ %d  lds %d" % (n, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))
' "$2"
