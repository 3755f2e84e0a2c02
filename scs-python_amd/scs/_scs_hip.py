"""scs._scs_hip — the MI355X (gfx950) backend module selected by
``LinearSolver.HIP_INDIRECT``.

It is the counterpart of one of the reference's per-backend CPython extension
modules (R:scs/scspy.c, R:scs/scsmodule.h, R:scs/scsobject.h — compiled once per
backend, R:meson.build:238-391).  Same module surface:

* ``version()``, ``sizeof_int()``, ``sizeof_float()``         (R:scs/scsmodule.h:16-23)
* type ``SCS(shape, Ax, Ai, Ap, Px, Pi, Pp, b, c, cone, **settings)`` with
  ``solve(warm_start, x, y, s)`` and ``update(b, c)``          (R:scs/scsobject.h:442-1225)

The reference glue is C against the CPython API; this one is Python over the
C-ABI of ``libscs_hip.so`` (include/scs_hip.h) via ctypes.  ctypes drops the GIL
for the duration of every foreign call, which reproduces the reference's
``Py_BEGIN_ALLOW_THREADS`` around scs_init/scs_solve/scs_update
(R:scs/scsobject.h:902-905,984-987,1216-1219); a per-instance lock protects
work/sol exactly as there (:892-899).

There is NO CPU fallback: if the shared library or a GPU is missing this module
raises (ImportError at import for a missing library; ValueError("ScsWork
allocation error!") from the constructor when no device is usable).
"""
import ctypes as C
import importlib.util
import os
import threading
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libscs_hip.so"

c_int, c_dbl = C.c_int, C.c_double
_PI, _PD = C.POINTER(c_int), C.POINTER(c_dbl)


# ---------------------------------------------------------------- C structs (include/scs_types.h)
class _ScsMatrix(C.Structure):
    _fields_ = [("x", _PD), ("i", _PI), ("p", _PI), ("m", c_int), ("n", c_int)]


class _ScsData(C.Structure):
    _fields_ = [("m", c_int), ("n", c_int), ("A", C.POINTER(_ScsMatrix)),
                ("P", C.POINTER(_ScsMatrix)), ("b", _PD), ("c", _PD)]


class _ScsCone(C.Structure):
    _fields_ = [("z", c_int), ("l", c_int), ("bu", _PD), ("bl", _PD), ("bsize", c_int),
                ("q", _PI), ("qsize", c_int), ("s", _PI), ("ssize", c_int),
                ("cs", _PI), ("cssize", c_int), ("ep", c_int), ("ed", c_int),
                ("p", _PD), ("psize", c_int)]


class _ScsSettings(C.Structure):
    _fields_ = [("normalize", c_int), ("scale", c_dbl), ("adaptive_scale", c_int),
                ("rho_x", c_dbl), ("max_iters", c_int), ("eps_abs", c_dbl),
                ("eps_rel", c_dbl), ("eps_infeas", c_dbl), ("alpha", c_dbl),
                ("time_limit_secs", c_dbl), ("verbose", c_int), ("warm_start", c_int),
                ("acceleration_lookback", c_int), ("acceleration_interval", c_int),
                ("acceleration_type_1", c_int), ("acceleration_regularization", c_dbl),
                ("acceleration_relaxation", c_dbl), ("write_data_filename", C.c_char_p),
                ("log_csv_filename", C.c_char_p)]


class _ScsSolution(C.Structure):
    _fields_ = [("x", _PD), ("y", _PD), ("s", _PD)]


class _ScsAaStats(C.Structure):
    _fields_ = [("iter", c_int), ("n_accept", c_int), ("n_reject_lapack", c_int),
                ("n_reject_rank0", c_int), ("n_reject_nonfinite", c_int),
                ("n_reject_weight_cap", c_int), ("n_safeguard_reject", c_int),
                ("last_rank", c_int), ("last_aa_norm", c_dbl), ("last_regularization", c_dbl)]


class _ScsInfo(C.Structure):
    _fields_ = [("iter", c_int), ("status", C.c_char * 128), ("lin_sys_solver", C.c_char * 128),
                ("status_val", c_int), ("scale_updates", c_int), ("pobj", c_dbl), ("dobj", c_dbl),
                ("res_pri", c_dbl), ("res_dual", c_dbl), ("gap", c_dbl), ("res_infeas", c_dbl),
                ("res_unbdd_a", c_dbl), ("res_unbdd_p", c_dbl), ("comp_slack", c_dbl),
                ("setup_time", c_dbl), ("solve_time", c_dbl), ("scale", c_dbl),
                ("lin_sys_time", c_dbl), ("cone_time", c_dbl), ("accel_time", c_dbl),
                ("rejected_accel_steps", c_int), ("accepted_accel_steps", c_int),
                ("aa_stats", _ScsAaStats), ("cg_iters", c_int)]


# ---------------------------------------------------------------- library loading
def _preload_hip_runtime():
    """Make sure exactly ONE HIP runtime ends up in the process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7,
    same as /opt/rocm's).  If libscs_hip.so pulled in the system copy first and
    torch were imported afterwards, torch would load its bundled copy too (it asks
    for the un-versioned file name) and the process would hold two runtimes.
    Preloading torch's copy (without importing torch) makes both resolve to it.
    SCS_HIP_RUNTIME=system skips this.
    """
    if os.environ.get("SCS_HIP_RUNTIME", "torch") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _runtime_env():
    """Runtime configuration that must be in the environment BEFORE the HIP runtime reads its flags (csrc/runtime.hpp
    scs_hip_runtime_env has the measurements): keep the runtime from pinning the caller's pageable arrays for host <->
    device copies — the driver evicts the process's queues for 30-80 ms some time after such pages are released.
    An existing value wins; SCS_HIP_RUNTIME_ENV=0 leaves the environment alone."""
    if os.environ.get("SCS_HIP_RUNTIME_ENV", "1") != "0":
        os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1000000")


def _load():
    _runtime_env()
    path = os.environ.get("SCS_HIP_LIB") or os.path.join(_HERE, _LIB_NAME)   # (SCS_HIP_LIB: an alternative build of the library, A/B labs)
    if not os.path.exists(path):
        raise ImportError(
            "scs._scs_hip: %s is not built (run `python __graft_entry__.py` / "
            "`make -C scs-python_amd`); this backend has no CPU fallback" % path)
    _preload_hip_runtime()
    try:
        lib = C.CDLL(path)
    except OSError as e:  # missing ROCm runtime etc.
        raise ImportError("scs._scs_hip: cannot load %s: %s" % (path, e))
    lib.scs_init.restype = C.c_void_p
    lib.scs_init.argtypes = [C.POINTER(_ScsData), C.POINTER(_ScsCone), C.POINTER(_ScsSettings)]
    lib.scs_hip_init_linsys.restype = C.c_void_p
    lib.scs_hip_init_linsys.argtypes = [C.POINTER(_ScsData), C.POINTER(_ScsCone), C.POINTER(_ScsSettings), c_int]
    lib.scs_hip_linsys_kind.restype = c_int
    lib.scs_hip_linsys_kind.argtypes = [C.c_void_p]
    lib.scs_hip_kkt_solve_dense.restype = c_int
    lib.scs_hip_kkt_solve_dense.argtypes = [C.POINTER(_ScsMatrix), C.POINTER(_ScsMatrix), _PD, _PD]
    lib.scs_solve.restype = c_int
    lib.scs_solve.argtypes = [C.c_void_p, C.POINTER(_ScsSolution), C.POINTER(_ScsInfo), c_int]
    lib.scs_hip_solve_batch.restype = c_int
    lib.scs_hip_solve_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.POINTER(_ScsSolution)),
                                        C.POINTER(C.POINTER(_ScsInfo)), c_int, c_int]
    lib.scs_update.restype = c_int
    lib.scs_update.argtypes = [C.c_void_p, _PD, _PD]
    lib.scs_finish.restype = None
    lib.scs_finish.argtypes = [C.c_void_p]
    lib.scs_set_default_settings.restype = None
    lib.scs_set_default_settings.argtypes = [C.POINTER(_ScsSettings)]
    lib.scs_version.restype = C.c_char_p
    lib.scs_sizeof_int.restype = C.c_size_t
    lib.scs_sizeof_float.restype = C.c_size_t
    lib.scs_hip_device_count.restype = c_int
    lib.scs_hip_set_device.restype = c_int
    lib.scs_hip_set_device.argtypes = [c_int]
    lib.scs_hip_labs_build.restype = c_int
    lib.scs_hip_mem_info.restype = c_int
    lib.scs_hip_mem_info.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.scs_hip_set_thread_device.restype = c_int
    lib.scs_hip_set_thread_device.argtypes = [c_int]
    lib.scs_hip_last_error.restype = C.c_char_p
    lib.scs_hip_spmv.restype = c_int
    lib.scs_hip_spmv.argtypes = [C.POINTER(_ScsMatrix), _PD, _PD, c_int]
    lib.scs_hip_cs_layout_host_spmv.restype = c_int
    lib.scs_hip_cs_layout_host_spmv.argtypes = [C.POINTER(_ScsMatrix), _PD, _PD, c_int, c_int, c_int]
    lib.scs_hip_spmv_bench.restype = c_dbl
    lib.scs_hip_spmv_bench.argtypes = [C.POINTER(_ScsMatrix), c_int, c_int]
    lib.scs_hip_proj_cone.restype = c_int
    lib.scs_hip_proj_cone.argtypes = [_PD, C.POINTER(_ScsCone), c_int, c_int]
    lib.scs_hip_proj_cone_seq.restype = c_int
    lib.scs_hip_proj_cone_seq.argtypes = [_PD, C.POINTER(_ScsCone), c_int, c_int, c_int, _PD, c_int]
    lib.scs_hip_kkt_solve.restype = c_int
    lib.scs_hip_kkt_solve.argtypes = [C.POINTER(_ScsMatrix), C.POINTER(_ScsMatrix), _PD, _PD, c_dbl, _PI]
    lib.scs_hip_normalize.restype = c_int
    lib.scs_hip_normalize.argtypes = [C.POINTER(_ScsMatrix), C.POINTER(_ScsMatrix), _PD, _PD,
                                      C.POINTER(_ScsCone), _PD, _PD, _PD]
    lib.scs_hip_set_profiling.restype = None
    lib.scs_hip_set_profiling.argtypes = [C.c_void_p, c_int]
    lib.scs_hip_kernel_times.restype = None
    lib.scs_hip_kernel_times.argtypes = [C.c_void_p, _PD]
    lib.scs_hip_solution_to_device.restype = c_int
    lib.scs_hip_solution_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.scs_hip_set_mark.restype = None
    lib.scs_hip_set_mark.argtypes = [C.c_void_p, c_int]
    lib.scs_hip_get_mark.restype = None
    lib.scs_hip_get_mark.argtypes = [C.c_void_p, _PD]
    lib.scs_hip_time_psd.restype = c_int
    lib.scs_hip_time_psd.argtypes = [C.c_void_p, c_int, _PD]
    lib.scs_hip_trim_pool.restype = None
    lib.scs_hip_trim_pool.argtypes = []
    lib.scs_hip_spin_fallbacks.restype = C.c_long
    lib.scs_hip_spin_fallbacks.argtypes = []
    lib.scs_hip_psd_refine_stats.restype = c_int
    lib.scs_hip_psd_refine_stats.argtypes = [C.c_void_p, _PD, c_int]
    lib.scs_hip_time_matvec.restype = c_int
    lib.scs_hip_time_matvec.argtypes = [C.c_void_p, c_int, _PD]
    lib.scs_hip_copy_bandwidth.restype = c_dbl
    lib.scs_hip_copy_bandwidth.argtypes = [C.c_size_t, c_int]
    lib.scs_hip_aa_init.restype = C.c_void_p
    lib.scs_hip_aa_init.argtypes = [c_int, c_int, c_int, c_dbl, c_dbl, c_dbl, c_dbl]
    lib.scs_hip_aa_apply.restype = c_dbl
    lib.scs_hip_aa_apply.argtypes = [C.c_void_p, _PD, _PD]
    lib.scs_hip_aa_safeguard.restype = c_int
    lib.scs_hip_aa_safeguard.argtypes = [C.c_void_p, _PD, _PD]
    lib.scs_hip_aa_reset.restype = None
    lib.scs_hip_aa_reset.argtypes = [C.c_void_p]
    lib.scs_hip_aa_get_stats.restype = None
    lib.scs_hip_aa_get_stats.argtypes = [C.c_void_p, C.POINTER(_ScsAaStats)]
    lib.scs_hip_aa_last_gamma.restype = c_int
    lib.scs_hip_aa_last_gamma.argtypes = [C.c_void_p, _PD]
    lib.scs_hip_aa_finish.restype = None
    lib.scs_hip_aa_finish.argtypes = [C.c_void_p]
    return lib


_lib = _load()


def version():
    return _lib.scs_version().decode()


def sizeof_int():
    return int(_lib.scs_sizeof_int())


def sizeof_float():
    return int(_lib.scs_sizeof_float())


def device_count():
    return int(_lib.scs_hip_device_count())


def labs_build():
    """True when the loaded library is the -DSCS_HIP_LABS build (csrc/options.hpp: the experiments that lost and the lab switches of
    the kernels read the environment there; the product does not contain them)"""
    return bool(_lib.scs_hip_labs_build())


def mem_info():
    """(free, total) HBM bytes of the device the next SCS(...) would use (blocks this library caches count as free); None without a device"""
    f, t = C.c_size_t(0), C.c_size_t(0)
    if _lib.scs_hip_mem_info(C.byref(f), C.byref(t)) != 0:
        return None
    return int(f.value), int(t.value)


def set_device(dev):
    if _lib.scs_hip_set_device(int(dev)) != 0:
        raise ValueError("invalid HIP device %r" % (dev,))


def set_thread_device(dev):
    """device of the calling THREAD's subsequent SCS(...) constructions (None / negative: back to the process default)"""
    if _lib.scs_hip_set_thread_device(-1 if dev is None else int(dev)) != 0:
        raise ValueError("invalid HIP device index %r" % (dev,))


def last_error():
    return _lib.scs_hip_last_error().decode()


# ---------------------------------------------------------------- argument parsing
def _pd(a):
    return a.ctypes.data_as(_PD)


def _pi(a):
    return a.ctypes.data_as(_PI)


def _float_array(name, a, length=None, exc_len=ValueError, len_msg=None):
    """1-D numpy float array of any float dtype -> contiguous float64 copy
    (R:scs/scsobject.h:60-68,574-592; integer arrays are rejected, test
    R:test/test_scs_coverage.py:1698-1720)."""
    if not isinstance(a, np.ndarray):
        raise TypeError("%s must be a 1-D numpy array of floats" % name)
    if not np.issubdtype(a.dtype, np.floating) or a.ndim != 1:
        raise TypeError("%s must be a 1-D numpy array of floats" % name)
    if length is not None and a.shape[0] != length:
        raise exc_len(len_msg or ("%s has incompatible dimension" % name))
    return np.array(a, dtype=np.float64, order="C", copy=True)


def _int_array(name, a):
    if not isinstance(a, np.ndarray):
        raise TypeError("%s must be a 1-D numpy array of ints" % name)
    if not np.issubdtype(a.dtype, np.integer) or a.ndim != 1:
        raise TypeError("%s must be a 1-D numpy array of ints" % name)
    if a.size and (a.max() > np.iinfo(np.int32).max or a.min() < np.iinfo(np.int32).min):
        raise ValueError("%s does not fit the 32-bit index type of the GPU backend" % name)
    return np.array(a, dtype=np.int32, order="C", copy=True)


def _as_c_int(name, v):
    """mirrors the 'i' format unit of PyArg_ParseTupleAndKeywords"""
    if isinstance(v, bool):
        return int(v)
    if isinstance(v, (float, np.floating)):
        raise TypeError("'float' object cannot be interpreted as an integer")
    try:
        iv = v.__index__()
    except AttributeError:
        raise TypeError("'%s' object cannot be interpreted as an integer" % type(v).__name__)
    if not -2 ** 31 <= iv < 2 ** 31:
        raise OverflowError("signed integer is greater than maximum")
    return int(iv)


def _as_c_double(name, v):
    """mirrors the 'd' format unit"""
    if isinstance(v, (str, bytes)) or v is None:
        raise TypeError("must be real number, not %s" % type(v).__name__)
    try:
        return float(v)
    except (TypeError, ValueError):
        raise TypeError("must be real number, not %s" % type(v).__name__)


def _as_bool(name, v):
    """mirrors 'O!' with PyBool_Type (R:scs/scsobject.h:534-536)"""
    if not isinstance(v, (bool, np.bool_)):
        raise TypeError("argument '%s' must be bool, not %s" % (name, type(v).__name__))
    return 1 if v else 0


def _as_filename(name, v):
    if v is None:
        return None
    if isinstance(v, str):
        return v.encode()
    raise TypeError("argument '%s' must be str or None, not %s" % (name, type(v).__name__))


_INT_SETTINGS = ("max_iters", "acceleration_lookback", "acceleration_interval", "acceleration_type_1")
_FLOAT_SETTINGS = ("scale", "eps_abs", "eps_rel", "eps_infeas", "alpha", "rho_x", "time_limit_secs",
                   "acceleration_regularization", "acceleration_relaxation")
_BOOL_SETTINGS = ("verbose", "normalize", "adaptive_scale")
_FILE_SETTINGS = ("write_data_filename", "log_csv_filename")


def _cone_err(key):
    return ValueError("Invalid value for cone field '%s'" % key)


def _cone_pos_int(cone, key):
    """R:scs/scsobject.h:86-127 — python ints only, non-negative, must fit scs_int"""
    if key not in cone:
        return 0
    v = cone[key]
    if isinstance(v, (bool, np.bool_)) or not isinstance(v, (int, np.integer)):
        raise _cone_err(key)
    v = int(v)
    if v < 0 or v >= 2 ** 31:
        raise _cone_err(key)
    return v


def _cone_int_list(cone, key):
    """list, bare int or numpy int array of non-negative ints (R:scs/scsobject.h:148-238)"""
    if key not in cone or cone[key] is None:
        return np.zeros(0, dtype=np.int32)
    v = cone[key]
    if isinstance(v, (int, np.integer)) and not isinstance(v, (bool, np.bool_)):
        v = [int(v)]
    if isinstance(v, np.ndarray):
        if v.ndim != 1 or not np.issubdtype(v.dtype, np.integer):
            raise _cone_err(key)
        vals = [int(t) for t in v]
    elif isinstance(v, (list, tuple)):
        vals = []
        for t in v:
            if isinstance(t, (bool, np.bool_)) or not isinstance(t, (int, np.integer)):
                raise _cone_err(key)
            vals.append(int(t))
    else:
        raise _cone_err(key)
    for t in vals:
        if t < 0 or t >= 2 ** 31:
            raise _cone_err(key)
    return np.asarray(vals, dtype=np.int32)


def _cone_float_list(cone, key):
    if key not in cone or cone[key] is None:
        return np.zeros(0, dtype=np.float64)
    v = cone[key]
    if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, (bool, np.bool_)):
        v = [float(v)]
    try:
        arr = np.asarray(v, dtype=np.float64)
    except (TypeError, ValueError):
        raise _cone_err(key)
    if arr.ndim != 1:
        raise _cone_err(key)
    return np.ascontiguousarray(arr)


def _info_dict(info):
    """ScsInfo -> the reference's info dict (R:scs/scsobject.h:1073-1111) + this backend's extras"""
    aa = info.aa_stats
    return {
        "status_val": int(info.status_val),
        "iter": int(info.iter),
        "scale_updates": int(info.scale_updates),
        "scale": float(info.scale),
        "pobj": float(info.pobj),
        "dobj": float(info.dobj),
        "res_pri": float(info.res_pri),
        "res_dual": float(info.res_dual),
        "gap": float(info.gap),
        "res_infeas": float(info.res_infeas),
        "res_unbdd_a": float(info.res_unbdd_a),
        "res_unbdd_p": float(info.res_unbdd_p),
        "comp_slack": float(info.comp_slack),
        "solve_time": float(info.solve_time),
        "setup_time": float(info.setup_time),
        "lin_sys_time": float(info.lin_sys_time),
        "cone_time": float(info.cone_time),
        "accel_time": float(info.accel_time),
        "rejected_accel_steps": int(info.rejected_accel_steps),
        "accepted_accel_steps": int(info.accepted_accel_steps),
        "status": info.status.decode(),
        "aa_stats": {
            "iter": int(aa.iter), "n_accept": int(aa.n_accept),
            "n_reject_lapack": int(aa.n_reject_lapack), "n_reject_rank0": int(aa.n_reject_rank0),
            "n_reject_nonfinite": int(aa.n_reject_nonfinite),
            "n_reject_weight_cap": int(aa.n_reject_weight_cap),
            "n_safeguard_reject": int(aa.n_safeguard_reject), "last_rank": int(aa.last_rank),
            "last_aa_norm": float(aa.last_aa_norm),
            "last_regularization": float(aa.last_regularization),
        },
        # extras of this backend (the reference's dict is a subset)
        "cg_iters": int(info.cg_iters),
        "lin_sys_solver": info.lin_sys_solver.decode(),
    }


class SCS(object):
    """Raw backend type; `scs.SCS` (scs/__init__.py) is the user-facing wrapper."""

    # linear-system solver this backend type builds (include/scs_hip.h scs_hip_init_linsys): the MODULE decides, as in the reference
    # (R:scs/py/__init__.py:40-66) — 1 = sparse indirect (PCG) here, scs._scs_hip_dense.SCS sets 2 (dense direct).  Only the bare
    # C entry scs_init consults SCS_HIP_LINSYS (LinearSolver.HIP_INDIRECT must not turn dense behind the caller's back).
    _LINSYS = 1

    def __init__(self, shape, Ax, Ai, Ap, Px, Pi, Pp, b, c, cone, **settings):
        if getattr(self, "_work", None):
            raise ValueError("Workspace already setup!")
        self._work = None
        self._lock = threading.Lock()
        # ---- shape "(ii)"
        if not isinstance(shape, tuple) or len(shape) != 2:
            raise TypeError("argument 1 must be 2-item sequence (m, n)")
        m, n = _as_c_int("m", shape[0]), _as_c_int("n", shape[1])
        if not isinstance(cone, dict):
            raise TypeError("argument 10 must be dict, not %s" % type(cone).__name__)
        # ---- settings (keyword table R:scs/scsobject.h:467-495)
        st = _ScsSettings()
        _lib.scs_set_default_settings(C.byref(st))
        self._fn_keep = []
        for key, val in settings.items():
            if key in _INT_SETTINGS:
                setattr(st, key, _as_c_int(key, val))
            elif key in _FLOAT_SETTINGS:
                setattr(st, key, _as_c_double(key, val))
            elif key in _BOOL_SETTINGS:
                setattr(st, key, _as_bool(key, val))
            elif key in _FILE_SETTINGS:
                fn = _as_filename(key, val)
                self._fn_keep.append(fn)
                setattr(st, key, fn)
            else:
                raise TypeError("'%s' is an invalid keyword argument for SCS()" % key)
        if m <= 0:
            raise ValueError("m must be a positive integer")
        if n <= 0:
            raise ValueError("n must be a positive integer")
        self.m, self.n = m, n
        # ---- data
        Axc = _float_array("Ax", Ax)
        Aic = _int_array("Ai", Ai)
        Apc = _int_array("Ap", Ap)
        if Apc.shape[0] != n + 1 or Axc.shape[0] != Aic.shape[0] or (Apc.size and Apc[-1] != Axc.shape[0]):
            raise ValueError("A has inconsistent CSC arrays")
        have_P = Px is not None and Pi is not None and Pp is not None
        if have_P:
            Pxc = _float_array("Px", Px)
            Pic = _int_array("Pi", Pi)
            Ppc = _int_array("Pp", Pp)
            if Ppc.shape[0] != n + 1 or Pxc.shape[0] != Pic.shape[0] or (Ppc.size and Ppc[-1] != Pxc.shape[0]):
                raise ValueError("P has inconsistent CSC arrays")
        cc = _float_array("c", c, n, ValueError, "c has incompatible dimension with A")
        bc = _float_array("b", b, m, ValueError, "b has incompatible dimension with A")
        # ---- cone (R:scs/scsobject.h:684-749)
        f_tmp = _cone_pos_int(cone, "f")
        z = _cone_pos_int(cone, "z")
        if f_tmp > 0:
            warnings.warn("The 'f' cone field is deprecated; use 'z' (Zero cone) instead. "
                          "If both 'f' and 'z' are set they are summed.", DeprecationWarning, stacklevel=2)
            z += f_tmp
        lcone = _cone_pos_int(cone, "l")
        bu = _cone_float_list(cone, "bu")
        bl = _cone_float_list(cone, "bl")
        if bu.size != bl.size:
            raise ValueError("bu different dimension to bl")
        q = _cone_int_list(cone, "q")
        s = _cone_int_list(cone, "s")
        cs = _cone_int_list(cone, "cs")
        p = _cone_float_list(cone, "p")
        ep = _cone_pos_int(cone, "ep")
        ed = _cone_pos_int(cone, "ed")
        # ---- settings validation (R:scs/scsobject.h:810-868)
        if st.max_iters <= 0:
            raise ValueError("max_iters must be positive")
        if st.acceleration_lookback < 0:
            raise ValueError("acceleration_lookback must be nonnegative (use acceleration_type_1=0 for type-II AA)")
        if st.acceleration_interval <= 0:
            raise ValueError("acceleration_interval must be positive")
        if not np.isfinite(st.acceleration_regularization) or st.acceleration_regularization < 0:
            raise ValueError("acceleration_regularization must be a nonnegative finite number")
        if (not np.isfinite(st.acceleration_relaxation) or st.acceleration_relaxation < 0
                or st.acceleration_relaxation > 2):
            raise ValueError("acceleration_relaxation must be in [0, 2]")
        if not np.isfinite(st.scale) or st.scale <= 0:
            raise ValueError("scale must be a positive finite number")
        if np.isnan(st.time_limit_secs) or st.time_limit_secs < 0:
            raise ValueError("time_limit_secs must be nonnegative")
        if np.isnan(st.eps_abs) or st.eps_abs < 0:
            raise ValueError("eps_abs must be nonnegative")
        if np.isnan(st.eps_rel) or st.eps_rel < 0:
            raise ValueError("eps_rel must be nonnegative")
        if np.isnan(st.eps_infeas) or st.eps_infeas < 0:
            raise ValueError("eps_infeas must be nonnegative")
        if not np.isfinite(st.alpha) or st.alpha <= 0 or st.alpha >= 2:
            raise ValueError("alpha must be in (0, 2)")
        if not np.isfinite(st.rho_x) or st.rho_x <= 0:
            raise ValueError("rho_x must be a positive finite number")
        st.warm_start = 0
        # ---- hand over to the C core (it copies everything, R:scs/scsobject.h:908)
        A = _ScsMatrix(_pd(Axc), _pi(Aic), _pi(Apc), m, n)
        P = _ScsMatrix(_pd(Pxc), _pi(Pic), _pi(Ppc), n, n) if have_P else None
        d = _ScsData(m, n, C.pointer(A), C.pointer(P) if have_P else None, _pd(bc), _pd(cc))
        k = _ScsCone(z, lcone, _pd(bu), _pd(bl), (bu.size + 1) if bu.size > 0 else 0,
                     _pi(q), q.size, _pi(s), s.size, _pi(cs), cs.size, ep, ed, _pd(p), p.size)
        self._x = np.zeros(n)
        self._y = np.zeros(m)
        self._s = np.zeros(m)
        if self._LINSYS:
            work = _lib.scs_hip_init_linsys(C.byref(d), C.byref(k), C.byref(st), self._LINSYS)
        else:
            work = _lib.scs_init(C.byref(d), C.byref(k), C.byref(st))  # GIL released by ctypes
        if not work:
            # the reference's message (R:scs/scsobject.h:903-912) + the backend's own reason: invalid data, no GPU,
            # out of HBM, or a limit this backend has and the reference has not (INTEGRATION.md "Limits")
            self._init_error = last_error()
            raise ValueError("ScsWork allocation error!" + (" (%s)" % self._init_error if self._init_error else ""))
        self._work = work

    # ------------------------------------------------------------------ solve
    def _warm(self, name, dst, src):
        """R:scs/scsobject.h:129-146"""
        if not isinstance(src, np.ndarray) or not np.issubdtype(src.dtype, np.floating) or src.ndim != 1:
            raise ValueError("Unable to parse %s warm-start" % name)
        if src.shape[0] != dst.shape[0]:
            raise ValueError("Unable to parse %s warm-start" % name)
        dst[:] = src

    def solve(self, warm_start=True, x=None, y=None, s=None):
        if not isinstance(warm_start, (bool, np.bool_)):
            raise TypeError("argument 1 must be bool, not %s" % type(warm_start).__name__)
        with self._lock:
            if not self._work:
                raise ValueError("Workspace not initialized!")
            if warm_start:
                if x is not None:
                    self._warm("x", self._x, x)
                if y is not None:
                    self._warm("y", self._y, y)
                if s is not None:
                    self._warm("s", self._s, s)
            sol = _ScsSolution(_pd(self._x), _pd(self._y), _pd(self._s))
            info = _ScsInfo()
            _lib.scs_solve(self._work, C.byref(sol), C.byref(info), 1 if warm_start else 0)
            # fresh copies owning their data (R:scs/scsobject.h:993-1043), taken under the lock
            xo, yo, so = self._x.copy(), self._y.copy(), self._s.copy()
        info_dict = _info_dict(info)
        return {"x": xo, "y": yo, "s": so, "info": info_dict}

    # ----------------------------------------------------------------- update
    def update(self, b=None, c=None):
        cc = bc = None
        if c is not None:
            if not isinstance(c, np.ndarray) or not np.issubdtype(c.dtype, np.floating) or c.ndim != 1:
                raise TypeError("c_new must be a 1-D numpy array of floats")
            if c.shape[0] != self.n:
                raise ValueError("c_new has incompatible dimension with A")
            cc = np.array(c, dtype=np.float64, order="C", copy=True)
        if b is not None:
            if not isinstance(b, np.ndarray) or not np.issubdtype(b.dtype, np.floating) or b.ndim != 1:
                raise TypeError("b_new must be a 1-D numpy array of floats")
            if b.shape[0] != self.m:
                raise ValueError("b_new has incompatible dimension with A")
            bc = np.array(b, dtype=np.float64, order="C", copy=True)
        with self._lock:
            if not self._work:
                raise ValueError("Workspace not initialized!")
            _lib.scs_update(self._work, _pd(bc) if bc is not None else None, _pd(cc) if cc is not None else None)
        return None

    # -------------------------------------------------- bench hooks (not part of the reference surface)
    def _set_profiling(self, on):
        with self._lock:
            _lib.scs_hip_set_profiling(self._work, 1 if on else 0)

    def _kernel_times(self):
        out = np.zeros(12)
        with self._lock:
            _lib.scs_hip_kernel_times(self._work, _pd(out))
        return {"k1_ms": out[0], "k1_n": int(out[1]), "k2_ms": out[2], "k2_n": int(out[3]),
                "nnz": int(out[4]), "k1_wgs": int(out[5]), "k2_wgs": int(out[6]), "nnz_p": int(out[7]),
                "cone_ms": out[8], "cone_n": int(out[9]), "k3_ms": out[10], "k3_n": int(out[11])}

    def solution_to_device(self, x_ptr=None, y_ptr=None, s_ptr=None):
        """copy the last solve's (x, y, s) from the workspace's HBM buffers to DEVICE addresses (ints, e.g.
        torch.Tensor.data_ptr() of float64 tensors on the same GPU); None skips a vector"""
        with self._lock:
            if not self._work:
                raise ValueError("Workspace not initialized!")
            rc = _lib.scs_hip_solution_to_device(self._work, x_ptr or None, y_ptr or None, s_ptr or None)
        if rc != 0:
            raise RuntimeError("libscs_hip: " + last_error())

    def _set_mark(self, it):
        _lib.scs_hip_set_mark(self._work, int(it))

    def _get_mark(self):
        out = np.zeros(4)
        _lib.scs_hip_get_mark(self._work, _pd(out))
        return {"ms": float(out[0]), "cg_iters": int(out[1]), "aa_calls": int(out[2]), "aa_accept": int(out[3])}

    def _time_psd(self, reps=20):
        out = np.zeros(4)
        with self._lock:
            rc = _lib.scs_hip_time_psd(self._work, int(reps), _pd(out))
        if rc == 1:
            return None
        _check(rc)
        return {"ms": float(out[0]), "matrices": int(out[1]), "max_order": int(out[2]), "ref_flops": float(out[3])}

    def _psd_refine_stats(self, cap=4096):
        """per PSD matrix of order > 32: [calls refined, refinements sent back to the sweeps, |K1|_F^2 at the last gate,
        mixed-sign off-norm^2 / |A|^2 after the last refinement, stage flag of the last call, then what the last call's matrix looked like
        as it arrived: |K1|_F^2, |off|^2 / |A|^2, omega] (include/scs_hip.h)"""
        out = np.zeros(8 * cap)
        with self._lock:
            cnt = _lib.scs_hip_psd_refine_stats(self._work, _pd(out), int(cap))
        if cnt < 0:
            raise RuntimeError("libscs_hip: " + last_error())
        return out[:8 * cnt].reshape(cnt, 8)

    def _time_matvec(self, reps=20):
        out = np.zeros(4)
        with self._lock:
            rc = _lib.scs_hip_time_matvec(self._work, int(reps), _pd(out))
        if rc != 0:
            raise RuntimeError("libscs_hip: " + last_error())
        return {"k1_ms": float(out[0]), "k2_ms": float(out[1]), "k3_ms": float(out[2]), "k3_back_to_back_ms": float(out[3])}

    def __del__(self):
        lock = getattr(self, "_lock", None)
        work = getattr(self, "_work", None)
        if work and lock is not None and _lib is not None:  # (_lib is None while the interpreter shuts down)
            with lock:
                _lib.scs_finish(self._work)
                self._work = None


def solve_batch(solvers, warm_start=False):
    """Grouped solve of several backend `SCS` objects (include/scs_hip.h: scs_hip_solve_batch): equally shaped
    problems share every kernel launch of the ADMM loop.  Returns the list of result dicts `solve()` would have
    returned for each (iterates bit-identical to separate solves).  warm_start uses each object's stored solution.
    A member that fails comes back with status "failure" like a failed .solve(); RuntimeError only for argument /
    whole-call errors."""
    if not isinstance(warm_start, (bool, np.bool_)):
        raise TypeError("argument 2 must be bool, not %s" % type(warm_start).__name__)
    solvers = list(solvers)
    if not solvers:
        return []
    if len(set(id(sv) for sv in solvers)) != len(solvers):
        raise ValueError("solve_batch: a solver appears twice")
    for sv in solvers:
        if not isinstance(sv, SCS):
            raise TypeError("solve_batch expects scs._scs_hip.SCS objects")
    ordered = sorted(solvers, key=id)  # one global lock order: two overlapping batches cannot deadlock
    for sv in ordered:
        sv._lock.acquire()
    try:
        for sv in solvers:
            if not sv._work:
                raise ValueError("Workspace not initialized!")
        cnt = len(solvers)
        sols = [_ScsSolution(_pd(sv._x), _pd(sv._y), _pd(sv._s)) for sv in solvers]
        infos = [_ScsInfo() for _ in solvers]
        works = (C.c_void_p * cnt)(*[sv._work for sv in solvers])
        solp = (C.POINTER(_ScsSolution) * cnt)(*[C.pointer(so) for so in sols])
        infp = (C.POINTER(_ScsInfo) * cnt)(*[C.pointer(io) for io in infos])
        rc = _lib.scs_hip_solve_batch(works, solp, infp, cnt, 1 if warm_start else 0)  # GIL released by ctypes
        err = last_error() if rc != 0 else ""
        out = [{"x": sv._x.copy(), "y": sv._y.copy(), "s": sv._s.copy(), "info": _info_dict(io)}
               for sv, io in zip(solvers, infos)]
    finally:
        for sv in ordered:
            sv._lock.release()
    # rc != 0 also when a single member ended SCS_FAILED: its dict says so (status_val -4, NaN vectors) exactly as .solve() would
    # report it, and the other members keep their results.  Only a call that failed as a whole raises.
    if rc != 0 and any(o["info"]["status_val"] == 0 for o in out):
        raise RuntimeError("libscs_hip: " + err)
    return out


# ---------------------------------------------------------------- kernel-level entry points (tests, bench)
def _matrix(A):
    """scipy CSC -> (_ScsMatrix, keepalive)"""
    x = np.ascontiguousarray(A.data, dtype=np.float64)
    i = np.ascontiguousarray(A.indices, dtype=np.int32)
    p = np.ascontiguousarray(A.indptr, dtype=np.int32)
    return _ScsMatrix(_pd(x), _pi(i), _pi(p), A.shape[0], A.shape[1]), (x, i, p)


def _cone_struct(cone):
    z = _cone_pos_int(cone, "z") + _cone_pos_int(cone, "f")
    bu, bl = _cone_float_list(cone, "bu"), _cone_float_list(cone, "bl")
    q, s, cs = _cone_int_list(cone, "q"), _cone_int_list(cone, "s"), _cone_int_list(cone, "cs")
    p = _cone_float_list(cone, "p")
    k = _ScsCone(z, _cone_pos_int(cone, "l"), _pd(bu), _pd(bl), (bu.size + 1) if bu.size else 0,
                 _pi(q), q.size, _pi(s), s.size, _pi(cs), cs.size,
                 _cone_pos_int(cone, "ep"), _cone_pos_int(cone, "ed"), _pd(p), p.size)
    return k, (bu, bl, q, s, cs, p)


def _check(rc):
    if rc != 0:
        raise RuntimeError("libscs_hip: " + last_error())


def spmv(A, x, transpose=False):
    """y = A x (or A' x) through the hot-path SpMV kernel; A scipy CSC."""
    M, keep = _matrix(A)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(A.shape[1] if transpose else A.shape[0])
    _check(_lib.scs_hip_spmv(C.byref(M), _pd(xx), _pd(y), 1 if transpose else 0))
    return y


def cs_layout_host_spmv(A, x, transpose=False, rpt=0, split=1):
    """HOST-ONLY: A x (or A' x) evaluated by walking the column-sorted pass layout of the large-matrix SpMV kernels
    as built by the host builder; None when the pattern does not fit the format.  (tests, no GPU needed)"""
    M, keep = _matrix(A)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(A.shape[1] if transpose else A.shape[0])
    rc = _lib.scs_hip_cs_layout_host_spmv(C.byref(M), _pd(xx), _pd(y), 1 if transpose else 0, int(rpt), int(split))
    if rc == 1:
        return None
    _check(rc)
    return y


def cs_layout_host_spmv_pieces(A, x, transpose=False, piece_len=24):
    """HOST-ONLY: the same walk for the virtual-row variant of the layout — rows too long for the count fields cut into pieces of
    at most piece_len nonzeros that ride in the passes, a row's pieces added as the device adds them; None when no row is that
    long or the pattern does not fit.  (tests, no GPU needed)"""
    M, keep = _matrix(A)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(A.shape[1] if transpose else A.shape[0])
    rc = _lib.scs_hip_cs_layout_host_spmv_pieces(C.byref(M), _pd(xx), _pd(y), 1 if transpose else 0, int(piece_len))
    if rc == 1:
        return None
    _check(rc)
    return y


def spmv_bench(A, transpose=False, reps=20):
    M, keep = _matrix(A)
    ms = _lib.scs_hip_spmv_bench(C.byref(M), 1 if transpose else 0, int(reps))
    if ms < 0:
        raise RuntimeError("libscs_hip: " + last_error())
    return ms


def proj_cone(z, cone, dual=False):
    x = np.array(z, dtype=np.float64, copy=True)
    k, keep = _cone_struct(cone)
    _check(_lib.scs_hip_proj_cone(_pd(x), C.byref(k), x.size, 1 if dual else 0))
    return x


def trim_pool():
    """return the library's cached device blocks to the driver (include/scs_hip.h scs_hip_trim_pool)"""
    _lib.scs_hip_trim_pool()


def spin_fallbacks():
    """solves of this process restarted without spinning multi-workgroup kernels after a barrier timed out (include/scs_hip.h)"""
    return int(_lib.scs_hip_spin_fallbacks())


def proj_cone_seq(zs, cone, dual=False, stats_cap=0):
    """rows of zs projected one after the other through one set of warm-started cone workspaces (include/scs_hip.h);
    returns (projections, refinement records of the large PSD matrices)"""
    xs = np.array(zs, dtype=np.float64, copy=True, order="C")
    assert xs.ndim == 2
    k, keep = _cone_struct(cone)
    st = np.zeros(8 * max(stats_cap, 1))
    cnt = _lib.scs_hip_proj_cone_seq(_pd(xs), C.byref(k), xs.shape[1], 1 if dual else 0, xs.shape[0], _pd(st), int(stats_cap))
    if cnt < 0:
        raise RuntimeError("libscs_hip: " + last_error())
    return xs, st[:8 * cnt].reshape(cnt, 8)


def kkt_solve(A, P, diag_r, rhs, tol=1e-12):
    M, keep = _matrix(A)
    Pm = None
    if P is not None:
        Pm, keep2 = _matrix(P)
    r = np.array(rhs, dtype=np.float64, copy=True)
    dr = np.ascontiguousarray(diag_r, dtype=np.float64)
    its = c_int(0)
    _check(_lib.scs_hip_kkt_solve(C.byref(M), C.byref(Pm) if Pm is not None else None, _pd(dr), _pd(r),
                                  float(tol), C.byref(its)))
    return r, its.value


def kkt_solve_dense(A, P, diag_r, rhs):
    """The same KKT system through the dense direct linsys (csrc/dense.hpp): explicit inverse of the reduced matrix."""
    M, keep = _matrix(A)
    Pm = None
    if P is not None:
        Pm, keep2 = _matrix(P)
    r = np.array(rhs, dtype=np.float64, copy=True)
    dr = np.ascontiguousarray(diag_r, dtype=np.float64)
    _check(_lib.scs_hip_kkt_solve_dense(C.byref(M), C.byref(Pm) if Pm is not None else None, _pd(dr), _pd(r)))
    return r


def normalize(A, P, b, c, cone):
    """Returns (A_data_hat, P_data_hat, b_hat, c_hat, D, E, sigma) as scs_init computes them."""
    M, keepA = _matrix(A.copy())
    Pm, keepP = (None, None)
    if P is not None:
        Pm, keepP = _matrix(P.copy())
    bb = np.array(b, dtype=np.float64, copy=True)
    cc = np.array(c, dtype=np.float64, copy=True)
    k, keep = _cone_struct(cone)
    D, E, sig = np.zeros(A.shape[0]), np.zeros(A.shape[1]), np.zeros(1)
    _check(_lib.scs_hip_normalize(C.byref(M), C.byref(Pm) if Pm is not None else None, _pd(bb), _pd(cc),
                                  C.byref(k), _pd(D), _pd(E), _pd(sig)))
    return keepA[0], (keepP[0] if keepP else None), bb, cc, D, E, float(sig[0])


def copy_bandwidth(nbytes=1 << 30, reps=10):
    return float(_lib.scs_hip_copy_bandwidth(int(nbytes), int(reps)))


_AA_STAT_FIELDS = ("iter", "n_accept", "n_reject_lapack", "n_reject_rank0", "n_reject_nonfinite",
                   "n_reject_weight_cap", "n_safeguard_reject", "last_rank", "last_aa_norm", "last_regularization")


class AndersonAccelerator(object):
    """The device Anderson accelerator of the ADMM loop as a standalone object on host vectors — the
    aa_init / aa_apply / aa_safeguard / aa_reset interface of the SCS core's aa.c (tests drive it next to the oracle)."""

    def __init__(self, dim, mem, type1=True, regularization=1e-8, relaxation=1.0, safeguard_factor=1.0,
                 max_weight_norm=1e10):
        self._h = _lib.scs_hip_aa_init(int(dim), int(mem), 1 if type1 else 0, float(regularization),
                                       float(relaxation), float(safeguard_factor), float(max_weight_norm))
        if not self._h:
            raise RuntimeError("libscs_hip: " + last_error())
        self.dim = int(dim)

    def apply(self, f, x):
        """f = F(x).  Returns (aa_norm, f_out): f_out is the extrapolated iterate when the step was accepted."""
        ff = np.array(f, dtype=np.float64, copy=True)
        xx = np.ascontiguousarray(x, dtype=np.float64)
        assert ff.size == self.dim and xx.size == self.dim
        nrm = _lib.scs_hip_aa_apply(self._h, _pd(ff), _pd(xx))
        if nrm != nrm and last_error():
            raise RuntimeError("libscs_hip: " + last_error())
        return nrm, ff

    def safeguard(self, f_new, x_new):
        """Returns (rc, f_new, x_new); rc = -1: rejected, the pair was rolled back to the pre-extrapolation one."""
        ff = np.array(f_new, dtype=np.float64, copy=True)
        xx = np.array(x_new, dtype=np.float64, copy=True)
        rc = _lib.scs_hip_aa_safeguard(self._h, _pd(ff), _pd(xx))
        if rc == -2:
            raise RuntimeError("libscs_hip: " + last_error())
        return rc, ff, xx

    def reset(self):
        _lib.scs_hip_aa_reset(self._h)

    def stats(self):
        st = _ScsAaStats()
        _lib.scs_hip_aa_get_stats(self._h, C.byref(st))
        return {k: getattr(st, k) for k in _AA_STAT_FIELDS}

    def last_gamma(self):
        n = _lib.scs_hip_aa_last_gamma(self._h, None)
        g = np.zeros(max(n, 1))
        _lib.scs_hip_aa_last_gamma(self._h, _pd(g))
        return g[:n]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.scs_hip_aa_finish(h)
