"""Compile spec_kernel.hip against a system's specialisation header with --save-temps and keep the gfx950 assembly.
  python tools/dump_spec_asm.py [puppet|puppet_basic|scissor_lift] [out.s] [extra flags...]"""
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from trep_amd import specialize, systems  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "puppet"
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/%s_spec.s" % name
    extra = sys.argv[3:]
    system = {"puppet": systems.puppet, "puppet_basic": systems.puppet_basic, "scissor_lift": lambda: systems.scissor_lift(4)}[name]()
    text = specialize.header(system)
    with tempfile.TemporaryDirectory() as tmp:
        hdr = os.path.join(tmp, "spec.hpp")
        open(hdr, "w").write(text)
        cmd = [specialize.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-I", specialize._CSRC,
               '-DTG_SPEC_HEADER="%s"' % hdr] + specialize._flags(text) + extra + ["--save-temps", "-o", os.path.join(tmp, "x.so"),
                                                                                  os.path.join(specialize._CSRC, "spec_kernel.hip")]
        subprocess.run(cmd, cwd=tmp, check=True)
        shutil.copy(os.path.join(tmp, "spec_kernel-hip-amdgcn-amd-amdhsa-gfx950.s"), out)
    print(out)


if __name__ == "__main__":
    main()
