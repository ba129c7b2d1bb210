"""System-specialised rollout kernels: the schedule of one ``System`` compiled into the kernel.

The generic kernels of ``libtrepamd.so`` interpret a flat schedule (``csrc/program.hpp``: ~110 integers and ~65 table
pointers per system) at run time.  For the long rollouts of the benchmark configurations that is where a large share
of the instructions goes: scalar registers cannot hold the schedule, so it is spilled into vector-register lanes and
re-read (``v_readlane``) at every use, and every LDS address is computed from run-time offsets.  ``build(system)``
asks the library for the specialisation header of the system (``tg_system_spec_header``: every integer a
``static constexpr``, every table a constant array), compiles ``csrc/spec_kernel.hip`` -- the SAME template source as
the generic kernel, instantiated on that header -- with hipcc for gfx950, and caches the result under
``trep_amd/_spec/`` keyed by the hash of header and sources.  ``BatchMidpointVI.specialize()`` then makes the batch's
rollouts use it.  Same arithmetic as the generic kernel: identical Newton iteration counts, states equal up to the
compiler's FMA-contraction choices (<= 1e-12 relative; the specialised build also uses its own scheduler flags).  The
library carries the FNV-1a hash of the header it was compiled against (``tg_spec_key``); ``tg_batch_load_specialized``
compares it with the batch's own system, also for a ``TREPAMD_SPEC_OVERRIDE`` library.

hipcc runs as a child process and needs no GPU, so specialisations can be built ahead of time (``__graft_entry__.build``
builds the BASELINE systems') and travel with the package; building on first use works too, as long as hipcc is there.
"""
import ctypes
import hashlib
import os
import subprocess

from . import _lib
from .descriptor import flatten

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
CACHE = os.path.join(_HERE, "_spec")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
_SOURCES = ["spec_kernel.hip", "mvi_core.hpp", "program.hpp", "bbd.hpp", "bbd_solve.hpp", "dual.hpp"]
# -amdgpu-mfma-vgpr-form: the matrix-core accumulators of the out-of-line solvers (gj_panel) stay in VGPRs.  As AGPRs they are ADDED to the
# rollout kernel's own 254 VGPRs (288 registers: ONE wave per SIMD instead of two -- 76 instead of 51 ms per benchmark launch)
DEFAULT_FLAGS = ("-DSPEC_ARGS_IN_MEMORY -DSPEC_DERIVATIVES -DTG_GJ_INLINE -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp "
                 "-mllvm -amdgpu-mfma-vgpr-form")


def header(system, with_key=False):
    """The generated C++ header for `system` (host-only: no GPU is touched); with_key: also its 64-bit hash."""
    L = _lib.lib()
    desc = flatten(system)
    h = L.tg_system_create(desc.byref())
    if not h:
        raise _lib.LibraryError(L.tg_last_error().decode())
    try:
        n = L.tg_system_spec_header(h, None, 0)
        if n <= 0:
            raise _lib.LibraryError(L.tg_last_error().decode())
        buf = ctypes.create_string_buffer(int(n))
        L.tg_system_spec_header(h, buf, n)
        text = buf.value.decode()
        return (text, int(L.tg_system_spec_key(h))) if with_key else text
    finally:
        L.tg_system_destroy(h)


def _flags(text=""):
    """Extra compiler flags of the specialised kernel (TREPAMD_SPEC_FLAGS overrides; part of the cache key).  A system whose
    team is a full wavefront gets helper waves in its second-derivative kernel (mvi_core.hpp, TG_HELPER_WAVES;
    TREPAMD_HELPER_WAVES=1 turns them off)."""
    flags = os.environ.get("TREPAMD_SPEC_FLAGS", DEFAULT_FLAGS).split()
    waves = int(os.environ.get("TREPAMD_HELPER_WAVES", "2"))
    if waves not in (1, 2):      # the schedule's pair lists are split in exactly two parts (csrc/program.hpp)
        raise ValueError("TREPAMD_HELPER_WAVES must be 1 or 2, not %d" % waves)
    if waves > 1 and "#define SPEC_TEAM 64\n" in text and not any(f.startswith("-DTG_HELPER_WAVES") for f in flags):
        flags.append("-DTG_HELPER_WAVES=%d" % waves)
    return flags


def _key(text):
    m = hashlib.sha256(text.encode())
    m.update(" ".join(_flags(text)).encode())
    for f in _SOURCES + [os.path.join("..", "..", "include", "trep_amd.h")]:
        with open(os.path.join(_CSRC, f), "rb") as fh:
            m.update(fh.read())
    return m.hexdigest()[:16]


def library_path(system, with_key=False):
    text, key = header(system, with_key=True)
    path = os.path.join(CACHE, "libtrepamd_spec_%s.so" % _key(text))
    return (path, text, key) if with_key else (path, text)


def build(system, force=False, verbose=False):
    """Path of the specialised kernel library of `system`, compiling it if it is not cached.  Safe to call from several
    processes at once (every rank of a multi-GPU launch does): each compiles against its own copy of the header and
    publishes header and library by atomic rename."""
    path, text, key = library_path(system, with_key=True)
    if os.path.exists(path) and not force:
        return path
    os.makedirs(CACHE, exist_ok=True)
    hdr = path[:-3] + ".hpp"
    mine = ".tmp%d" % os.getpid()
    with open(hdr + mine, "w") as fh:
        fh.write(text)
    tmp = path + mine
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-I", _CSRC, '-DTG_SPEC_HEADER="%s"' % (hdr + mine), "-DTG_SPEC_KEY=0x%016xull" % key] + _flags(text) + \
          ["-o", tmp, os.path.join(_CSRC, "spec_kernel.hip")]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    if r.returncode != 0:
        try:
            os.remove(hdr + mine)
        except OSError:
            pass
        raise _lib.LibraryError("specialisation failed:\n" + r.stderr[-4000:])
    os.replace(hdr + mine, hdr)      # kept next to the library for inspection
    os.replace(tmp, path)
    return path


def is_built(system):
    return os.path.exists(library_path(system)[0])


def prune(systems_to_keep, verbose=False):
    """Remove every cached specialisation (library + header) under ``trep_amd/_spec/`` whose key none of `systems_to_keep` produces with
    the current sources and flags: a source edit changes every key, and the stale libraries would otherwise travel with the package
    (they are harmless -- ``tg_batch_load_specialized`` checks the key -- but they were 23 MB at the end of round 4).  Returns the
    removed file names."""
    if not os.path.isdir(CACHE):
        return []
    keep = set()
    for system in systems_to_keep:
        base = os.path.basename(library_path(system)[0])[:-3]
        keep.update((base + ".so", base + ".hpp"))
    gone = []
    for name in sorted(os.listdir(CACHE)):
        if name not in keep and ".tmp" not in name:
            try:
                os.remove(os.path.join(CACHE, name))
                gone.append(name)
            except OSError:
                pass
    if verbose:
        print("specialize.prune: removed %d stale files, kept %d" % (len(gone), len(keep)))
    return gone
