#!/usr/bin/env python3
"""Static instruction counts of the specialised rollout kernel attributed to SOURCE LINES (from a -gline-tables-only assembly:
python tools/dump_spec_asm.py puppet /tmp/puppet_g.s -gline-tables-only).  Groups the lines of mvi_core.hpp by the function /
phase ranges given below and prints VALU fp64 / VALU other / SALU / LDS / other per group; --lines prints the top lines.
    python tools/isa_by_line.py /tmp/puppet_g.s [--kernel 'k_specILi0ELi0'] [--lines 40]"""
import collections
import re
import sys

path = sys.argv[1]
kernel = "k_specILi0ELi0"
top = 0
min_depth = 0
if "--depth" in sys.argv:
    min_depth = int(sys.argv[sys.argv.index("--depth") + 1])
depth = 0
if "--kernel" in sys.argv:
    kernel = sys.argv[sys.argv.index("--kernel") + 1]
if "--lines" in sys.argv:
    top = int(sys.argv[sys.argv.index("--lines") + 1])
files = {}
cur = None
inside = False
counts = collections.defaultdict(lambda: collections.Counter())
seq = []
F64 = re.compile(r"^v_(fma|mul|add|fmac|max|min|rcp|ldexp|frexp|fract|rndne|trunc|cvt.*f64|cmp.*f64|div|sqrt|rsq).*_f64|^v_.*_f64")
for ln in open(path):
    t = ln.strip()
    m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", t)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if t.startswith("_Z") and t.split(":")[0].endswith(tuple("E")) and ":" in t:
        inside = kernel in t
        continue
    if t.startswith(".end_amdhsa_kernel") or t.startswith(".Lfunc_end"):
        inside = False if t.startswith(".Lfunc_end") else inside
    if not inside:
        continue
    md = re.search(r"Depth=(\d+)", ln)
    if md and ";" in ln and ("Loop Header" in ln or "in Loop" in ln or "Parent Loop" in ln or "Inner Loop" in ln):
        depth = int(md.group(1))
    if re.match(r"^\.LBB\d+_\d+:", ln) and "Depth" not in ln:
        depth = 0 if "in Loop" not in ln else depth
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        continue
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    if depth < min_depth:
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cls = "fp64" if "_f64" in op else ("dpp/lane" if ("dpp" in t or "readlane" in op or "writelane" in op or "readfirstlane" in op or "permlane" in op) else ("cndmask" if "cndmask" in op else ("mov" if op.startswith("v_mov") or "accvgpr" in op else "valu_int")))
    elif op.startswith("s_"):
        cls = "wait" if op in ("s_waitcnt", "s_nop") else ("branch" if "branch" in op else "salu")
    elif op.startswith("ds_"):
        cls = "lds"
    else:
        cls = "mem"
    counts[cur][cls] += 1
    seq.append((cur, cls))

GROUPS = []
if "--ranges" in sys.argv:
    for spec in sys.argv[sys.argv.index("--ranges") + 1].split(","):
        name, a, b = spec.split(":")
        GROUPS.append((name, int(a), int(b)))
cols = ["fp64", "valu_int", "cndmask", "mov", "dpp/lane", "salu", "branch", "wait", "lds", "mem"]
if GROUPS:
    agg = collections.defaultdict(collections.Counter)
    if "--inherit" in sys.argv:
        # helper lines (fma() in the HIP math header, the DPP / 16-byte access helpers, ...) count for the phase whose line came last in
        # program order: `seq` holds (location, class) in the order the instructions stand in the text
        cur_g = "prologue"
        for (f, l), cls in seq:
            g = None if f != "mvi_core.hpp" else next((n for n, a, b in GROUPS if a <= l <= b), None)
            if g is not None:
                cur_g = g
            agg[cur_g][cls] += 1
    else:
        for (f, l), c in counts.items():
            g = "other files" if f != "mvi_core.hpp" else next((n for n, a, b in GROUPS if a <= l <= b), "mvi_core other")
            agg[g].update(c)
    print("%-34s" % "group" + "".join("%9s" % c for c in cols) + "    total")
    for g, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        print("%-34s" % g + "".join("%9d" % c[k] for k in cols) + "%9d" % sum(c.values()))
tot = collections.Counter()
for c in counts.values():
    tot.update(c)
print("%-34s" % "TOTAL" + "".join("%9d" % tot[k] for k in cols) + "%9d" % sum(tot.values()))
if top:
    print()
    for (f, l), c in sorted(counts.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
        print("%-16s:%5d " % (f, l) + "".join("%9d" % c[k] for k in cols) + "%9d" % sum(c.values()))
