"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/trep_amd.h declares,
compiles a system descriptor (host-only call) and refuses to compute without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from common import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "trep_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from trep_amd import _lib
    L = _lib.lib()
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    assert set(_lib.exported_symbols()) <= set(names)
    assert b"gfx950" in L.tg_version()


def test_system_compiles_on_host_and_reports_schedule():
    from trep_amd import _lib
    L = _lib.lib()
    system, d = build("puppet40")
    h = L.tg_system_create(d.byref())
    assert h
    sizes = np.zeros(6, dtype=np.int32)
    assert L.tg_system_sizes(h, sizes.ctypes.data_as(_lib._c_ip)) == 0
    assert list(sizes) == [40, 22, 18, 0, 6, 80]
    info = np.zeros(8, dtype=np.int32)
    assert L.tg_system_info(h, info.ctypes.data_as(_lib._c_ip)) == 0
    team, lds, joints, levels, bodies, items, pairs, dh = (int(x) for x in info)
    assert team == 64 and joints == 34 and bodies == 10 and items == 88 and pairs == 442
    assert lds <= 20480          # 8 trajectories (waves) per CU out of 160 KiB LDS
    L.tg_system_destroy(h)


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly (this container has no GPU)."""
    from trep_amd import _lib
    import trep_amd
    if _lib.lib().tg_device_count() > 0:
        pytest.skip("a GPU is present")
    system, d = build("pendulum1")
    with pytest.raises(_lib.LibraryError):
        trep_amd.MidpointVI(system)
    with pytest.raises(_lib.LibraryError):
        trep_amd.BatchMidpointVI(system, 4)


def test_product_never_imports_oracle():
    """The oracle and the host emulation are test infrastructure: nothing under trep_amd/ may load them."""
    banned = ("import oracle", "from oracle", "libtreporacle", "libtrepamd_emu", "emu_harness")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "trep_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for b in banned:
                    assert b not in text, (f, b)


def test_specialisation_emitter_is_in_sync_and_header_is_complete():
    """csrc/spec_emit.inc is generated from struct DevProg (tools/gen_spec_emitter.py); the header it prints for a system
    must define every field of the struct as a compile-time constant (host-only: no GPU involved)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_spec_emitter.py"), "--check"]).returncode == 0, \
        "trep_amd/csrc/spec_emit.inc is stale: run tools/gen_spec_emitter.py"
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_spec_emitter
    from trep_amd import specialize, systems
    ints, dbl_arrays, pointers = gen_spec_emitter.parse_fields()
    text = specialize.header(systems.scissor_lift(4))
    for f in ints:
        assert "static constexpr int %s = " % f in text, f
    for ctype, name in pointers:
        assert "static constexpr const %s *%s = " % (ctype, name) in text, name
    assert "#define SPEC_TEAM 64" in text and "static constexpr int nc = 8;" in text
    # a different system gives a different header (and cache key)
    assert specialize.library_path(systems.pend_on_cart())[0] != specialize.library_path(systems.scissor_lift(4))[0]
