"""the reference's OWN CPython glue, read where it lies (`/root/reference/scs/scspy.c` with
`scsmodule.h` / `scsobject.h`), compiles against this repo's `include/` (glbopts.h, scs.h, scs_types.h) and links
against `libscs_hip.so` — unchanged.  The build goes to a temporary directory (nothing of the reference enters the
repo or travels to the GPU box); skipped where `/root/reference` does not exist.

The only test-local piece is a backport header for three CPython 3.13 functions the glue takes from the
`pythoncapi-compat` submodule (empty in the snapshot; `R:scs/scsobject.h:1,112,154,168`) — this image has CPython 3.10.
Without a GPU `scs_init` returns NULL, so what can run here is everything the glue does BEFORE the core is entered
plus the failure path: module surface, argument parsing and its messages, and "ScsWork allocation error!".
"""
import importlib.util
import os
import subprocess
import sys
import sysconfig

import numpy as np
import pytest
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/scs"
LIBDIR = os.path.join(ROOT, "scs-python_amd", "scs")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "scspy.c")),
                                reason="the reference tree is only present in the build container")

COMPAT = r"""
#include <Python.h>
#if PY_VERSION_HEX < 0x030D0000
static inline int PyDict_GetItemStringRef(PyObject *d, const char *k, PyObject **out) {
  PyObject *key = PyUnicode_FromString(k);
  if (!key) { *out = NULL; return -1; }
  PyObject *v = PyDict_GetItemWithError(d, key);
  Py_DECREF(key);
  if (v) { Py_INCREF(v); *out = v; return 1; }
  *out = NULL;
  return PyErr_Occurred() ? -1 : 0;
}
static inline PyObject *PyList_GetItemRef(PyObject *l, Py_ssize_t i) {
  PyObject *v = PyList_GetItem(l, i);
  Py_XINCREF(v);
  return v;
}
#endif
"""


def _cc_args(tmp):
    os.makedirs(os.path.join(tmp, "pythoncapi-compat"), exist_ok=True)
    with open(os.path.join(tmp, "pythoncapi-compat", "pythoncapi_compat.h"), "w") as f:
        f.write(COMPAT)
    return ["-I", os.path.join(ROOT, "include"), "-idirafter", str(tmp),
            "-I", sysconfig.get_paths()["include"], "-I", np.get_include(),
            "-DCTRLC=1", "-DCOPYAMATRIX=1"]      # the defines every backend target passes, R:meson.build:288-313


def test_reference_glue_compiles_against_include(tmp_path):
    """`gcc -fsyntax-only` of the glue: every type, field, macro and function it names exists in include/."""
    r = subprocess.run(["gcc", "-fsyntax-only", "-Werror=implicit-function-declaration", os.path.join(REF, "scspy.c")]
                       + _cc_args(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.fixture(scope="module")
def glue(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("refglue")
    so = os.path.join(tmp, "_scs_direct" + sysconfig.get_config_var("EXT_SUFFIX"))   # default module name, R:scs/scsmodule.h:98-99
    cmd = (["gcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(REF, "scspy.c")] + _cc_args(tmp)
           + ["-L", LIBDIR, "-lscs_hip", "-Wl,-rpath," + LIBDIR])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    from scs import _scs_hip  # noqa: F401  (preloads the HIP runtime the library links against)
    spec = importlib.util.spec_from_file_location("_scs_direct", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _args(m=2, n=1):
    A = sparse.csc_matrix(np.array([[1.0], [-1.0]]))
    return [(m, n), A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32), None, None, None,
            np.array([1.0, 0.0]), np.array([1.0]), {"l": 2}]


def test_reference_glue_links_and_reports_this_library(glue):
    import scs
    assert glue.version() == scs.__version__          # scs_version() of libscs_hip.so through R:scs/scsmodule.h:5
    assert glue.sizeof_int() == 4 and glue.sizeof_float() == 8
    assert hasattr(glue, "SCS")


def test_reference_glue_argument_checks_run_before_the_core(glue):
    a = _args()
    a[7] = np.array([1.0, 0.0, 3.0])
    with pytest.raises(ValueError, match="b has incompatible dimension with A"):      # R:scs/scsobject.h:675
        glue.SCS(*a, verbose=False)
    a = _args()
    a[9] = {"l": 2, "q": [-1]}
    with pytest.raises(ValueError):                                                     # parse_pos_scs_int, :169
        glue.SCS(*a, verbose=False)
    with pytest.raises(TypeError):                                                      # PyArg table, :498-551
        glue.SCS(*_args(), max_iters="many")


def test_reference_glue_init_reaches_scs_init(glue):
    """With a device this constructs a workspace (the GPU suite drives the same C-ABI through ctypes and the plain-C
    consumer tests/cabi/cabi_smoke.c); without one scs_init returns NULL and the glue reports it: R:scs/scsobject.h:903-914."""
    from scs import _scs_hip
    if _scs_hip.device_count() > 0:
        s = glue.SCS(*_args(), verbose=False, eps_abs=1e-9, eps_rel=1e-9)
        sol = s.solve(False, None, None, None)
        assert sol["info"]["status"] == "solved" and abs(sol["x"][0]) < 1e-6
    else:
        with pytest.raises(ValueError, match="ScsWork allocation error"):
            glue.SCS(*_args(), verbose=False)
