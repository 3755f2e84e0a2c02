"""CPU tests (no GPU) of the drop-in boundary:
 * the C-ABI library loads and exports every symbol include/scs_hip.h declares;
 * struct layouts used by the ctypes binding match the header;
 * the Python layer reproduces the reference's checks and messages (R:scs/py/__init__.py:102-166)
   and the raw backend type reproduces the glue's argument validation (R:scs/scsobject.h:442-914),
   all of which happen BEFORE the core is entered — so they run without a device;
 * on a box without a GPU the backend fails loudly (no CPU fallback).
"""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest
from scipy import sparse

import scs
from scs import _scs_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = _scs_hip.device_count() == 0


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "scs_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = re.findall(r"\b(scs_[a-z_0-9]+)\s*\(", hdr)
    assert {"scs_init", "scs_solve", "scs_update", "scs_finish", "scs_set_default_settings", "scs_version"} <= set(names)
    lib = ctypes.CDLL(os.path.join(ROOT, "scs-python_amd", "scs", "libscs_hip.so"))
    for n in names:
        assert hasattr(lib, n), "libscs_hip.so does not export %s" % n


def test_module_surface_matches_reference_backend_modules():
    # R:scs/scsmodule.h:16-23 / R:scs/py/__init__.py:9-11 / R:test/test_scs_coverage.py:787-797
    assert isinstance(_scs_hip.version(), str) and _scs_hip.version() == scs.__version__
    assert _scs_hip.sizeof_int() == scs.__sizeof_int__ == 4      # int32-only GPU build, R:meson.build:172-174
    assert _scs_hip.sizeof_float() == scs.__sizeof_float__ == 8
    assert hasattr(_scs_hip, "SCS")
    assert (scs.SOLVED, scs.SOLVED_INACCURATE, scs.UNBOUNDED, scs.INFEASIBLE, scs.FAILED) == (1, 2, -1, -2, -4)


def test_default_settings_through_the_c_abi():
    st = _scs_hip._ScsSettings()
    _scs_hip._lib.scs_set_default_settings(ctypes.byref(st))
    # R:README.md:98-104 and R:test/test_warm_start_consistency.py:228-241
    assert (st.acceleration_lookback, st.acceleration_interval, st.acceleration_type_1) == (10, 10, 1)
    assert st.acceleration_regularization == 1e-8 and st.acceleration_relaxation == 1.0
    assert (st.scale, st.rho_x, st.alpha) == (0.1, 1e-6, 1.5)
    assert (st.eps_abs, st.eps_rel, st.max_iters, st.normalize, st.adaptive_scale) == (1e-4, 1e-4, 100000, 1, 1)


def test_linear_solver_enum_and_dispatch():
    assert scs.LinearSolver("hip_indirect") is scs.LinearSolver.HIP_INDIRECT
    for name in ("AUTO", "QDLDL", "CPU_INDIRECT", "MKL", "ACCELERATE", "CPU_DENSE", "GPU_INDIRECT", "CUDSS"):
        assert hasattr(scs.LinearSolver, name)                      # R:scs/py/__init__.py:28-37
    stg = {"linear_solver": "hip_indirect", "verbose": False}
    assert scs._select_scs_module(stg) is _scs_hip and "linear_solver" not in stg   # popped, R:scs/py/__init__.py:71
    assert scs._select_scs_module({}) is _scs_hip                    # AUTO
    with pytest.raises(ImportError):                                  # un-built optional backend
        scs._select_scs_module({"linear_solver": scs.LinearSolver.CUDSS})
    with pytest.raises(ValueError):
        scs._select_scs_module({"linear_solver": "nope"})


A = sparse.csc_matrix(np.array([[1.0], [-1.0]]))
DATA = {"A": A, "b": np.array([1.0, 0.0]), "c": np.array([-1.0])}
CONE = {"l": 2}


@pytest.mark.parametrize("data,cone,exc,match", [
    ({"b": DATA["b"], "c": DATA["c"]}, CONE, ValueError, "Missing A"),
    ({"A": A, "c": DATA["c"]}, CONE, ValueError, "Missing one of b, c"),
    ({"A": A, "b": None, "c": DATA["c"]}, CONE, ValueError, "Incomplete data"),
    ({"A": np.eye(2), "b": DATA["b"], "c": np.ones(2)}, CONE, TypeError, "sparse"),
    ({"A": A, "b": np.ones(3), "c": DATA["c"]}, CONE, ValueError, "shape"),
    (dict(DATA, P=np.eye(1)), CONE, TypeError, "sparse"),
    (dict(DATA, P=sparse.eye(2, format="csc")), CONE, ValueError, "shape"),
    ({}, CONE, ValueError, "Missing data or cone"),
    (DATA, {}, ValueError, "Missing data or cone"),
])
def test_python_layer_checks(data, cone, exc, match):
    # R:test/test_scs_coverage.py:45-108 / R:scs/py/__init__.py:102-153
    with pytest.raises(exc, match=match):
        scs.SCS(data, cone, verbose=False)


def _expect_core_entered(fn):
    """the call passed every boundary check; without a GPU the core then refuses loudly"""
    if NO_GPU:
        with pytest.raises(ValueError, match="ScsWork allocation error!"):
            fn()
        assert "no HIP device" in _scs_hip.last_error()
    else:
        fn()


def test_csc_conversion_warns_and_never_mutates_caller():
    # R:test/test_scs_coverage.py:116-150 ("CSC" in the warning), R:scs/py/__init__.py:137-141
    Acsr = sparse.csr_matrix(np.array([[1.0], [-1.0]]))
    with pytest.warns(UserWarning, match="CSC"):
        _expect_core_entered(lambda: scs.SCS({"A": Acsr, "b": DATA["b"], "c": DATA["c"]}, CONE, verbose=False))
    assert Acsr.format == "csr"
    unsorted = sparse.csc_matrix((np.array([1.0, 2.0]), np.array([1, 0]), np.array([0, 2])), shape=(2, 1))
    unsorted.has_sorted_indices = False
    before = unsorted.indices.copy()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _expect_core_entered(lambda: scs.SCS({"A": unsorted, "b": DATA["b"], "c": DATA["c"]}, CONE, verbose=False))
    np.testing.assert_array_equal(unsorted.indices, before)


def test_upper_triangle_extraction_helper():
    P = sparse.csc_matrix(np.array([[2.0, 1.0], [1.0, 3.0]]))
    assert scs._has_lower_tri(P)
    assert not scs._has_lower_tri(sparse.triu(P, format="csc"))
    assert not scs._has_lower_tri(sparse.csc_matrix((2, 2)))


RAW = ((2, 1), A.data, A.indices, A.indptr, None, None, None, DATA["b"], DATA["c"], CONE)


def _raw(**kw):
    return _scs_hip.SCS(*RAW, verbose=False, **kw)


@pytest.mark.parametrize("kw,exc,match", [
    (dict(max_iters=0), ValueError, "max_iters must be positive"),
    (dict(max_iters=-5), ValueError, "max_iters must be positive"),
    (dict(max_iters=1.1), TypeError, "integer"),
    (dict(scale=0.0), ValueError, "scale must be"),
    (dict(scale=float("inf")), ValueError, "scale must be"),
    (dict(alpha=2.0), ValueError, r"alpha must be in \(0, 2\)"),
    (dict(alpha=float("nan")), ValueError, "alpha"),
    (dict(rho_x=-1.0), ValueError, "rho_x must be"),
    (dict(acceleration_interval=0), ValueError, "acceleration_interval must be positive"),
    (dict(acceleration_lookback=-1), ValueError, "acceleration_lookback must be nonnegative"),
    (dict(acceleration_relaxation=2.5), ValueError, "acceleration_relaxation"),
    (dict(acceleration_regularization=-1e-3), ValueError, "acceleration_regularization"),
    (dict(eps_abs=-1.0), ValueError, "eps_abs"),
    (dict(eps_rel=float("nan")), ValueError, "eps_rel"),
    (dict(eps_infeas=-1.0), ValueError, "eps_infeas"),
    (dict(time_limit_secs=-1.0), ValueError, "time_limit_secs"),
    (dict(eps_abs="tight"), TypeError, "real number"),
    (dict(normalize=1), TypeError, "bool"),
    (dict(not_a_setting=3), TypeError, "invalid keyword"),
])
def test_settings_validation(kw, exc, match):
    # R:test/test_scs_coverage.py:1094-1132,2324-2405 / R:scs/scsobject.h:810-868
    with pytest.raises(exc, match=match):
        _raw(**kw)


def test_inf_accepted_for_eps_and_time_limit():
    # R:test/test_scs_coverage.py:2366-2405: +inf is legal there
    _expect_core_entered(lambda: _raw(eps_abs=float("inf"), eps_rel=float("inf"), time_limit_secs=float("inf")))


@pytest.mark.parametrize("idx,bad,exc,match", [
    (1, A.data.astype(np.int64), TypeError, "Ax must be a 1-D numpy array of floats"),
    (1, list(A.data), TypeError, "numpy array"),
    (2, A.indices.astype(np.float64), TypeError, "Ai must be a 1-D numpy array of ints"),
    (7, np.array([1, 0]), TypeError, "b must be a 1-D numpy array of floats"),
    (7, np.ones(3), ValueError, "b has incompatible dimension"),
    (8, np.ones((1, 1)), TypeError, "c must be a 1-D numpy array of floats"),
    (9, [1, 2], TypeError, "dict"),
])
def test_array_argument_validation(idx, bad, exc, match):
    # R:test/test_scs_coverage.py:1193-1218,1698-1720 / R:scs/scsobject.h:574-683
    args = list(RAW)
    args[idx] = bad
    with pytest.raises(exc, match=match):
        _scs_hip.SCS(*args, verbose=False)


def test_float32_inputs_are_cast():
    # R:test/test_scs_coverage.py:2937-2956
    args = list(RAW)
    args[1] = A.data.astype(np.float32)
    args[7] = DATA["b"].astype(np.float32)
    _expect_core_entered(lambda: _scs_hip.SCS(*args, verbose=False))


@pytest.mark.parametrize("cone,match", [
    ({"l": 2, "q": [-1]}, "Invalid value for cone field 'q'"),
    ({"l": 2, "s": [1.5]}, "Invalid value for cone field 's'"),
    ({"l": -2}, "Invalid value for cone field 'l'"),
    ({"l": 2, "q": "abc"}, "Invalid value for cone field 'q'"),
    ({"l": 1, "bu": [1.0, 2.0], "bl": [0.0]}, "bu different dimension"),
])
def test_cone_parsing_errors(cone, match):
    # R:test/test_scs_coverage.py:2554,2567,2666 / R:scs/scsobject.h:74-80,718-721
    args = list(RAW)
    args[9] = cone
    with pytest.raises(ValueError, match=match):
        _scs_hip.SCS(*args, verbose=False)


@pytest.mark.parametrize("cone", [{"l": 2, "q": []}, {"l": np.int64(2)}, {"q": 2}, {"q": np.array([2])}, {"q": [2], "l": 0}])
def test_cone_value_forms_accepted(cone):
    # list, bare int, numpy int array (R:test/test_scs_coverage.py:2493-2533)
    args = list(RAW)
    args[9] = cone
    _expect_core_entered(lambda: _scs_hip.SCS(*args, verbose=False))


def test_f_cone_field_deprecated_and_summed():
    # R:test/test_scs_coverage.py:2448-2487 / R:scs/scsobject.h:692-704
    args = list(RAW)
    args[9] = {"f": 1, "l": 1}
    with pytest.warns(DeprecationWarning, match="'f' cone field"):
        _expect_core_entered(lambda: _scs_hip.SCS(*args, verbose=False))
    with warnings.catch_warnings():
        warnings.simplefilter("error", DeprecationWarning)
        with pytest.raises(DeprecationWarning):
            _scs_hip.SCS(*args, verbose=False)


def test_cone_dimension_mismatch_is_a_core_failure():
    # R:test/test_scs_basic.py:99-100,113-114 -> ValueError("ScsWork allocation error!")
    with pytest.raises(ValueError, match="ScsWork allocation error!"):
        scs.solve(DATA, {"q": [4]}, verbose=False)
    with pytest.raises(ValueError):
        scs.solve(DATA, {"q": [4], "l": -2})
    with pytest.raises(TypeError):
        scs.solve()


@pytest.mark.skipif(not NO_GPU, reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback_exists():
    with pytest.raises(ValueError, match="ScsWork allocation error!"):
        scs.SCS(DATA, CONE, verbose=False)
    assert "no CPU fallback" in _scs_hip.last_error()
    with pytest.raises(RuntimeError, match="no HIP device"):
        _scs_hip.spmv(A, np.ones(1))
    with pytest.raises(RuntimeError, match="no HIP device"):
        _scs_hip.proj_cone(np.ones(2), {"l": 2})
    # and the product package never imports the oracle
    import sys
    pkg = os.path.join(ROOT, "scs-python_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "scs_oracle" not in src and "liboscs" not in src and "oracle/" not in src.replace("the oracle", ""), f


def test_backend_modules_bind_their_linear_solver():
    """The module decides the linear solver, as in the reference (R:scs/py/__init__.py:40-66): scs._scs_hip binds the sparse indirect
    solver (1) whatever SCS_HIP_LINSYS says — only the bare C scs_init consults the environment — and scs._scs_hip_dense binds 2."""
    from scs import _scs_hip, _scs_hip_dense
    assert _scs_hip.SCS._LINSYS == 1
    assert _scs_hip_dense.SCS._LINSYS == 2


def test_product_library_reads_only_the_documented_environment_variables():
    """the product has at most 20 runtime knobs, all documented.  Every `SCS_HIP_*` name that occurs in
    libscs_hip.so (csrc/options.hpp parses them all in one place) is in INTEGRATION.md's "Runtime knobs" table; the switches of the
    experiments that lost exist in the -DSCS_HIP_LABS build only."""
    import re
    import subprocess
    from scs import _scs_hip
    lib = os.path.join(ROOT, "scs-python_amd", "scs", "libscs_hip.so")
    names = set(re.findall(rb"SCS_HIP_[A-Z0-9_]+", open(lib, "rb").read()))
    names = {n.decode() for n in names} - {"SCS_HIP_"}
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = text[text.index("## Runtime knobs"):text.index("**The labs build.**")]
    documented = set(re.findall(r"^\| `(SCS_HIP_[A-Z0-9_]+)`", table, flags=re.M))
    assert len(documented) <= 20, sorted(documented)
    assert names <= documented, sorted(names - documented)
    assert {"SCS_HIP_KRYLOV", "SCS_HIP_K1DOT", "SCS_HIP_PERSIST", "SCS_HIP_GRAPH", "SCS_HIP_PSD_COOP", "SCS_HIP_CS_SCHED"}.isdisjoint(names)
    if not os.environ.get("SCS_HIP_LIB"):
        assert not _scs_hip.labs_build()
