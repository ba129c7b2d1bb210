#!/usr/bin/env python3
"""Static opcode-class histogram of the system-specialised rollout kernel k_spec<0, 0> (and its out-of-line solvers): recompiles
spec_kernel.hip against the puppet's specialisation header with --save-temps and classifies every instruction of the kernel's text,
split by loop nesting depth (the assembler comments `in Loop: ... Depth=N`: depth 1 = the step loop, 2 = the Newton loop, >= 3 = loops
inside a phase).  The dynamic counts (what the SQ counters saw per DEL step) are in the rNN_sq_counters.json / rNN_valu_classes.json
files; this is the map of WHERE in the text the non-fp64 instructions sit.
    python tools/isa_histogram.py [system] > profiles/rNN_isa_histogram.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trep_amd import specialize, systems  # noqa: E402

CLASSES = [
    ("fp64 math", r"^v_(fma|fmac|mul|add|rcp|rsq|sqrt|max|min|trunc|rndne|floor|ceil|fract|ldexp|frexp\w*|div_\w+|cvt_f64_\w+|cvt_\w+_f64)_f64|^v_cvt_f64|^v_(mfma)_f64"),
    ("fp64 compare", r"^v_cmp\w*_f64|^v_cmpx\w*_f64"),
    ("fp32 math / cvt", r"^v_\w+_f32|^v_cvt_"),
    ("v_mov / v_accvgpr", r"^v_mov_b(32|64)|^v_accvgpr|^v_swap"),
    ("v_cndmask", r"^v_cndmask"),
    ("v_readlane / writelane / readfirstlane", r"^v_readlane|^v_writelane|^v_readfirstlane"),
    ("dpp (mov / fmac)", r"_dpp"),
    ("int add / sub / shift / logic (32)", r"^v_(add|sub|subrev|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|lshlrev|lshrrev|ashrrev|add3|lshl_add|lshl_or|and_or|or3|xad|min_u|max_u|min_i|max_i|mbcnt)\w*(_u32|_i32|_b32|_u16|_i16|_b16)|^v_mbcnt|^v_(add|sub)_co"),
    ("int mul / mad (32)", r"^v_mul_(lo|hi|u32|i32)|^v_mad_(u32|i32|u64|i64)|^v_mul_u32|^v_mad_u"),
    ("int 64-bit address arithmetic", r"^v_lshl_add_u64|^v_lshlrev_b64|^v_add_u64|^v_lshl_b64|^v_mad_u64"),
    ("int compare", r"^v_cmp\w*_(u32|i32|u64|i64|u16|i16|b32)|^v_cmpx"),
    ("ds_read", r"^ds_read|^ds_load"),
    ("ds_write", r"^ds_write|^ds_store"),
    ("ds atomic / other", r"^ds_"),
    ("global / flat / scratch load", r"^(global|flat|scratch|buffer)_load"),
    ("global / flat / scratch store, atomic", r"^(global|flat|scratch|buffer)_"),
    ("s_waitcnt / s_nop", r"^s_waitcnt|^s_nop|^s_sleep"),
    ("s_branch / s_cbranch", r"^s_c?branch|^s_setpc|^s_swappc|^s_call"),
    ("exec mask (saveexec, and/or exec)", r"saveexec|exec"),
    ("s_load / s_buffer_load", r"^s_load|^s_buffer_load|^s_memtime"),
    ("other SALU", r"^s_"),
    ("other VALU", r"^v_"),
]


def classify(op, text):
    if "_dpp" in text.split()[0]:
        return "dpp (mov / fmac)"
    for name, pat in CLASSES:
        if name.startswith("exec mask"):
            if re.search(r"saveexec", op) or re.search(r"\bexec\b", text):
                return name
            continue
        if re.search(pat, op):
            return name
    return "other"


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "puppet"
    system = {"puppet": systems.puppet, "puppet_basic": systems.puppet_basic, "scissor_lift": lambda: systems.scissor_lift(4)}[name]()
    text = specialize.header(system)
    with tempfile.TemporaryDirectory() as tmp:
        hdr = os.path.join(tmp, "spec.hpp")
        open(hdr, "w").write(text)
        cmd = [specialize.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-I", specialize._CSRC,
               '-DTG_SPEC_HEADER="%s"' % hdr] + specialize._flags(text) + ["--save-temps", "-o", os.path.join(tmp, "x.so"), os.path.join(specialize._CSRC, "spec_kernel.hip")]
        subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        asm = open(os.path.join(tmp, "spec_kernel-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    funcs = collections.OrderedDict()
    cur, depth = None, 0
    for line in asm:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, depth = m.group(1), 0
            funcs[cur] = collections.defaultdict(lambda: [0, 0, 0, 0])
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.strip().startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.search(r"Depth=(\d+)", line)
        if m and (";" in line):
            if "Loop Header" in line or "in Loop" in line or "Parent Loop" in line or "Inner Loop" in line:
                depth = int(m.group(1))
        if re.match(r"^\.LBB\d+_\d+:", line) and "Depth" not in line:
            depth = 0 if "in Loop" not in line else depth
        s = line.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        funcs[cur][classify(op, s)][min(depth, 3)] += 1

    def label(fn):
        m = re.search(r"k_specILi(\d)ELi(\d)", fn)
        if m:
            return "k_spec<%s, %s>" % (m.group(1), m.group(2))
        for key in ("gj_bbd", "gj_panel_rhs", "gj_panel", "gj_rows_exact", "gj_rows", "pivot_exact", "k_spec_debug_solve"):
            if key in fn:
                return key
        return fn[:40]
    print("# static opcode classes of the %s specialisation (flags: %s)" % (name, " ".join(specialize._flags(text))))
    print("# columns: instructions outside any loop | loop depth 1 (step loop) | depth 2 (Newton loop) | depth >= 3 (loops inside a phase)")
    for fn, table in funcs.items():
        lab = label(fn)
        if not (lab.startswith("k_spec<0, 0>") or lab in ("gj_bbd", "gj_panel")):
            continue
        tot = [sum(v[i] for v in table.values()) for i in range(4)]
        print("\n%s  -- %d instructions (%d / %d / %d / %d)" % (lab, sum(tot), *tot))
        for cname, v in sorted(table.items(), key=lambda kv: -sum(kv[1])):
            print("  %-44s %6d   (%5d %5d %5d %5d)" % (cname, sum(v), *v))


if __name__ == "__main__":
    main()
