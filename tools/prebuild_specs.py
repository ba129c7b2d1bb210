#!/usr/bin/env python3
"""Build the system-specialised kernels of every test system here (hipcc child processes, no GPU needed) so that a GPU-box run does
not spend box time compiling: python tools/prebuild_specs.py [workers]"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import BUILDERS          # noqa: E402
from trep_amd import specialize      # noqa: E402

systems = [make() for make in BUILDERS.values()]
with ThreadPoolExecutor(max_workers=int(sys.argv[1]) if len(sys.argv) > 1 else 6) as pool:
    paths = list(pool.map(specialize.build, systems))
print("%d specialisations under %s" % (len(set(paths)), specialize.CACHE))
