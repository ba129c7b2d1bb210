#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of the system-specialised libraries: recompiles spec_kernel.hip against the
specialisation header of each BASELINE system with the flags trep_amd/specialize.py uses plus
-Rpass-analysis=kernel-resource-usage, and prints one line per kernel.
    python tools/spec_resources.py [system ...] [-- extra flags]  >  profiles/rNN_resource_usage.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trep_amd import specialize, systems  # noqa: E402

NAMES = {"pendulum": lambda: systems.pendulum(1), "pend_on_cart": systems.pend_on_cart, "scissor_lift": lambda: systems.scissor_lift(4),
         "puppet": systems.puppet, "puppet_basic": systems.puppet_basic}
MODES = {"0": "rollout", "3": "deriv1", "4": "deriv2z"}


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        extra = args[args.index("--") + 1:]
        args = args[:args.index("--")]
    for name in (args or list(NAMES)):
        text = specialize.header(NAMES[name]())
        with tempfile.TemporaryDirectory() as tmp:
            hdr = os.path.join(tmp, "spec.hpp")
            open(hdr, "w").write(text)
            cmd = [specialize.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-I", specialize._CSRC,
                   '-DTG_SPEC_HEADER="%s"' % hdr] + specialize._flags(text) + extra + ["-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "x.so"),
                   os.path.join(specialize._CSRC, "spec_kernel.hip")]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if r.returncode != 0:
            print(name, "compile failed\n", r.stdout[-2000:])
            continue
        print("# %s   flags: %s" % (name, " ".join(specialize._flags(text) + extra)))
        print("%-34s %5s %5s %6s %6s %8s %5s %7s" % ("kernel", "VGPR", "AGPR", "SGPRsp", "VGPRsp", "scratchB", "occ", "LDS B"))
        cur = None
        rows = []
        for line in r.stdout.splitlines():
            m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
            if not m:
                continue
            body = m.group(1).strip()
            if body.startswith("Function Name:"):
                cur = {"name": body.split(":", 1)[1].strip()}
                rows.append(cur)
            elif cur is not None and ":" in body:
                k, v = body.split(":", 1)
                cur[k.strip()] = v.strip()
        for row in rows:
            nm = row["name"]
            m = re.search(r"k_specILi(\d)ELi(\d)", nm)
            if m:
                label = "k_spec<%s, %s> (%s%s)" % (m.group(1), m.group(2), MODES.get(m.group(1), "?"), ", exact pivot rule" if m.group(2) == "1" else "")
            elif "debug_solve" in nm:
                label = "k_spec_debug_solve (test hook)"
            else:
                label = nm[:34]
            print("%-34s %5s %5s %6s %6s %8s %5s %7s" % (label, row.get("VGPRs", "?"), row.get("AGPRs", "?"), row.get("SGPRs Spill", "?"), row.get("VGPRs Spill", "?"),
                                                       row.get("ScratchSize [bytes/lane]", "?"), row.get("Occupancy [waves/SIMD]", "?"), row.get("LDS Size [bytes/block]", "?")))
        print()


if __name__ == "__main__":
    main()
