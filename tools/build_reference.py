#!/usr/bin/env python3
"""Build an importable Python-3 working copy of the read-only reference in /tmp.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference
exists); never on the GPU box.  Nothing it produces is committed: the output
tree lives in /tmp/trep_ref (override with --out).  The only use of the built
reference is tools/gen_golden.py, which imports it to emit the golden
input/output vectors committed under tests/golden/.

The reference (MurpheyLab/trep) is Python-2 + a CPython-2 C extension.  The
port is purely mechanical:
  * lib2to3 over the Python layer, a handful of numpy-alias renames,
  * CPython-3 type-object / module-init / capsule spellings in the C layer.
No arithmetic is touched.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import sysconfig

REF = "/root/reference"

C_SOURCES = """midpointvi system math-code frame _trep config potential force input
constraint frametransform spline tapemeasure constraints/distance constraints/plane
constraints/point potentials/gravity potentials/linearspring potentials/configspring
potentials/nonlinear_config_spring forces/damping forces/lineardamper forces/configforce
forces/bodywrench forces/hybridwrench forces/spatialwrench forces/pistonexample""".split()


def sub_file(path, subs, count_required=True):
    with open(path) as fh:
        txt = fh.read()
    for pat, rep in subs:
        txt, n = re.subn(pat, rep, txt, flags=re.M)
        if count_required and n == 0:
            raise RuntimeError("pattern %r not found in %s" % (pat, path))
    with open(path, "w") as fh:
        fh.write(txt)


def port_python(root):
    pkg = os.path.join(root, "trep")
    targets = []
    for sub in ["", "discopt", "constraints", "potentials", "forces", "puppets"]:
        d = os.path.join(pkg, sub)
        targets += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".py")]
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n"] + targets,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(os.path.join(pkg, "__version__.py"), "w") as fh:
        fh.write("__version__ = 'reference-py3-port'\n")
    sub_file(os.path.join(pkg, "__init__.py"),
             [(r"^from __version__ import", "from .__version__ import")], False)
    for f in targets:
        sub_file(f, [(r"\bnp\.int\b", "np.int64"), (r"\bnp\.float\b", "np.float64"),
                     (r"\bnp\.object\b", "object")], False)
    # The puppet factory imports OpenGL / trep.visual at module scope; neither is
    # installed here and neither is on the hot path.
    sub_file(os.path.join(pkg, "puppets", "puppets.py"),
             [(r"^from OpenGL\.GL import \*\nfrom OpenGL\.GLU import \*\nfrom trep\.visual import \*\n",
               "try:\n    from OpenGL.GL import *\n    from OpenGL.GLU import *\n"
               "    from trep.visual import *\nexcept Exception:\n"
               "    class VisualItem3D(object): pass\n")])


def port_c(root):
    cdir = os.path.join(root, "trep", "_trep")
    for dirpath, _, files in os.walk(cdir):
        for f in files:
            if not f.endswith((".c", ".h")):
                continue
            sub_file(os.path.join(dirpath, f), [
                (r"PyObject_HEAD_INIT\(NULL\)\s*\n\s*0,\s*/\*\s*ob_size\s*\*/", "PyVarObject_HEAD_INIT(NULL, 0)"),
                (r"\b(\w+)->ob_type\b", r"Py_TYPE(\1)"),
                (r"\bPyInt_FromLong\b", "PyLong_FromLong"),
                (r"\bPyString_FromString\b", "PyUnicode_FromString"),
                (r"\bPyExc_StandardError\b", "PyExc_Exception"),
            ], False)
    sub_file(os.path.join(cdir, "c_api.h"), [
        (r"PyCObject_Check\(", "PyCapsule_CheckExact("),
        (r"PyCObject_AsVoidPtr\((\w+)\)", r'PyCapsule_GetPointer(\1, "trep._C_API")'),
        (r"PyCObject_FromVoidPtr\(([^,]+), NULL\)", r'PyCapsule_New(\1, "trep._C_API", NULL)'),
    ])
    sub_file(os.path.join(cdir, "_trep.c"), [
        (r"#ifndef PyMODINIT_FUNC", "#if 0"),
        (r"PyMODINIT_FUNC init_trep\(void\)",
         'static struct PyModuleDef trepmodule = {PyModuleDef_HEAD_INIT, "_trep", "trep C core", -1, CTrepMethods};\n'
         "PyMODINIT_FUNC PyInit__trep(void)"),
        (r"m = Py_InitModule3\(\"_trep\", CTrepMethods,\s*\n[^\n]*\);", "m = PyModule_Create(&trepmodule);"),
        (r"^(\s+)return;", r"\1return NULL;"),
    ])
    # the module-init function must return the module object
    path = os.path.join(cdir, "_trep.c")
    txt = open(path).read()
    idx = txt.rstrip().rfind("}")
    txt = txt[:idx] + "    return m;\n}\n"
    open(path, "w").write(txt)


def build(root):
    import numpy
    cdir = os.path.join(root, "trep", "_trep")
    inc = ["-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include()]
    objs = []
    for s in C_SOURCES:
        o = os.path.join(root, "obj_" + s.replace("/", "_") + ".o")
        subprocess.run(["gcc", "-O2", "-fPIC", "-fno-strict-aliasing", "-w"] + inc +
                       ["-c", os.path.join(cdir, s + ".c"), "-o", o], check=True)
        objs.append(o)
    so = os.path.join(root, "trep", "_trep" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.run(["gcc", "-shared"] + objs + ["-lpthread", "-lm", "-o", so], check=True)
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/trep_ref")
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not present (this tool only runs in the build container)")
    marker = os.path.join(args.out, ".built")
    if os.path.exists(marker) and not args.force:
        print("already built:", args.out)
        return
    if os.path.exists(args.out):
        shutil.rmtree(args.out)
    os.makedirs(args.out)
    shutil.copytree(os.path.join(REF, "trep"), os.path.join(args.out, "trep"))
    subprocess.run(["chmod", "-R", "u+w", args.out], check=True)
    port_python(args.out)
    port_c(args.out)
    so = build(args.out)
    open(marker, "w").write(so + "\n")
    print("built", so)


if __name__ == "__main__":
    main()
