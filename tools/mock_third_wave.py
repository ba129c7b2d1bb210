#!/usr/bin/env python3
"""TIMING MOCKS for the third wavefront per SIMD of the puppet's rollout kernel (round-4 verdict, item 3).  Never loaded by the package:
the libraries go to tools/ab/ and are only reachable through TREPAMD_SPEC_OVERRIDE.

A third wave per SIMD needs <= 13 653 B of LDS per trajectory (12 workgroups per CU) and <= 168 registers.  The real kernel has 20 128 B and
246.  What would it buy?  Built here, all with -DTG_MOCK_TIMING (three Newton iterations per step whatever the numbers say, guards of the
structured solve ignored, iterate frozen: the instruction stream of the real kernel on numbers that may be garbage):

  mock       the real LDS layout, 246 registers                      -> how faithful the mock's timing is (compare with the real kernel)
  mock_w3    the real LDS layout, compiled for 3 waves (168 registers, ~80 spilled): still 2 waves per SIMD -> the cost of the spills alone
  mock_lds   the pose / Newton-image union ALIASED onto the J / W areas: 13 632 B per trajectory, 246 registers: still 2 waves per SIMD
             (by registers) -> the LDS side alone (expected: nothing)
  mock_lds_w3  both: 12 workgroups per CU = 3 waves per SIMD, with the spill code

    python tools/mock_third_wave.py        (build box; then tools/mock_third_wave.sh runs them on one GPU box)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trep_amd import specialize, systems     # noqa: E402

OUT = os.path.join(ROOT, "tools", "ab")
UNION = ("o_Df", "o_sc", "o_G", "o_gB", "o_pE", "o_nE", "o_wR", "o_dqi")


def alias_union(text):
    """The header with the pose / image union moved onto the J area and everything behind it pulled down."""
    vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"static constexpr int (\w+) = (-?\d+);", text)}
    base, size = vals["o_Df"], vals["nf"] * vals["df_ld"]
    end = base + size
    shift = size

    def repl(m):
        name, v = m.group(1), int(m.group(2))
        if name in UNION:
            v = vals["o_J"] + (v - base)
        elif name.startswith("o_") and v >= end:
            v -= shift
        elif name == "lds_per_team":
            v -= shift
        return "static constexpr int %s = %d;" % (name, v)
    return re.sub(r"static constexpr int (\w+) = (-?\d+);", repl, text), vals["lds_per_team"] - shift


def build(name, text, key, extra):
    os.makedirs(OUT, exist_ok=True)
    hdr = os.path.join(OUT, "hdr_%s.hpp" % name)
    open(hdr, "w").write(text)
    lib = os.path.join(OUT, "lib_%s.so" % name)
    cmd = [specialize.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-I", specialize._CSRC,
           '-DTG_SPEC_HEADER="%s"' % hdr, "-DTG_SPEC_KEY=0x%016xull" % key] + specialize.DEFAULT_FLAGS.split() + extra + \
          ["-Rpass-analysis=kernel-resource-usage", "-o", lib, os.path.join(specialize._CSRC, "spec_kernel.hip")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-3000:])
    # resource usage of k_spec<0, 0>
    blocks = r.stdout.split("Function Name: ")
    for b in blocks:
        if "k_specILi0ELi0" in b.split("\n")[0]:
            g = lambda k: re.search(k + r": (\d+)", b).group(1)
            print("%-12s VGPRs %s  spilled %s  scratch %s B/lane  occupancy %s  (flags: %s)" % (name, g("VGPRs"), g("VGPRs Spill"), g("ScratchSize \[bytes/lane\]"), g("Occupancy \[waves/SIMD\]"), " ".join(extra)))
    return lib


def main():
    system = systems.puppet()
    text, key = specialize.header(system, with_key=True)
    aliased, lds = alias_union(text)
    print("aliased layout: %d doubles = %d B per trajectory (%d workgroups per CU)" % (lds, 8 * lds, (160 * 1024) // (8 * lds)))
    build("real", text, key, [])
    build("mock", text, key, ["-DTG_MOCK_TIMING"])
    build("mock_w3", text, key, ["-DTG_MOCK_TIMING", "-DTG_ROLLOUT_WAVES=3"])
    real = int(re.search(r"static constexpr int lds_per_team = (\d+);", text).group(1))
    build("mock_lds", aliased, key, ["-DTG_MOCK_TIMING", "-DTG_MOCK_REAL_LDS=%d" % real])
    build("mock_lds_w3", aliased, key, ["-DTG_MOCK_TIMING", "-DTG_ROLLOUT_WAVES=3", "-DTG_MOCK_REAL_LDS=%d" % real])


if __name__ == "__main__":
    main()
