"""ctypes front-end of the CPU ORACLE (oracle/liboscs.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (scs-python_amd/scs) never does.

The call surface mirrors the reference's raw extension type
(R:scs/scsobject.h:442-1131: `SCS(shape, Ax, Ai, Ap, Px, Pi, Pp, b, c, cone,
**settings)`, `.solve(warm_start, x, y, s)`, `.update(b, c)`) so that parity
tests can drive oracle and product with the same arguments.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSCS_LIB=omp (set before the first call, bench.py's all-core CPU leg only): the OpenMP TIMING build of the same
# sources — element-wise loops / mat-vecs on all cores, tree reductions; the tests always use the sequential checker
_OMP = os.environ.get("OSCS_LIB", "") == "omp"
_LIB_PATH = os.path.join(_HERE, "liboscs_omp.so" if _OMP else "liboscs.so")

c_int, c_dbl = C.c_int, C.c_double
PI, PD = C.POINTER(c_int), C.POINTER(c_dbl)


class ScsMatrix(C.Structure):
    _fields_ = [("x", PD), ("i", PI), ("p", PI), ("m", c_int), ("n", c_int)]


class ScsData(C.Structure):
    _fields_ = [("m", c_int), ("n", c_int), ("A", C.POINTER(ScsMatrix)),
                ("P", C.POINTER(ScsMatrix)), ("b", PD), ("c", PD)]


class ScsCone(C.Structure):
    _fields_ = [("z", c_int), ("l", c_int), ("bu", PD), ("bl", PD), ("bsize", c_int),
                ("q", PI), ("qsize", c_int), ("s", PI), ("ssize", c_int),
                ("cs", PI), ("cssize", c_int), ("ep", c_int), ("ed", c_int),
                ("p", PD), ("psize", c_int)]


class ScsSettings(C.Structure):
    _fields_ = [("normalize", c_int), ("scale", c_dbl), ("adaptive_scale", c_int),
                ("rho_x", c_dbl), ("max_iters", c_int), ("eps_abs", c_dbl),
                ("eps_rel", c_dbl), ("eps_infeas", c_dbl), ("alpha", c_dbl),
                ("time_limit_secs", c_dbl), ("verbose", c_int), ("warm_start", c_int),
                ("acceleration_lookback", c_int), ("acceleration_interval", c_int),
                ("acceleration_type_1", c_int), ("acceleration_regularization", c_dbl),
                ("acceleration_relaxation", c_dbl), ("write_data_filename", C.c_char_p),
                ("log_csv_filename", C.c_char_p)]


class ScsSolution(C.Structure):
    _fields_ = [("x", PD), ("y", PD), ("s", PD)]


class ScsAaStats(C.Structure):
    _fields_ = [("iter", c_int), ("n_accept", c_int), ("n_reject_lapack", c_int),
                ("n_reject_rank0", c_int), ("n_reject_nonfinite", c_int),
                ("n_reject_weight_cap", c_int), ("n_safeguard_reject", c_int),
                ("last_rank", c_int), ("last_aa_norm", c_dbl), ("last_regularization", c_dbl)]


class ScsInfo(C.Structure):
    _fields_ = [("iter", c_int), ("status", C.c_char * 128), ("lin_sys_solver", C.c_char * 128),
                ("status_val", c_int), ("scale_updates", c_int), ("pobj", c_dbl), ("dobj", c_dbl),
                ("res_pri", c_dbl), ("res_dual", c_dbl), ("gap", c_dbl), ("res_infeas", c_dbl),
                ("res_unbdd_a", c_dbl), ("res_unbdd_p", c_dbl), ("comp_slack", c_dbl),
                ("setup_time", c_dbl), ("solve_time", c_dbl), ("scale", c_dbl),
                ("lin_sys_time", c_dbl), ("cone_time", c_dbl), ("accel_time", c_dbl),
                ("rejected_accel_steps", c_int), ("accepted_accel_steps", c_int),
                ("aa_stats", ScsAaStats), ("cg_iters", c_int)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oscs_init.restype = C.c_void_p
        L.oscs_init.argtypes = [C.POINTER(ScsData), C.POINTER(ScsCone), C.POINTER(ScsSettings), c_int]
        L.oscs_solve.restype = c_int
        L.oscs_solve.argtypes = [C.c_void_p, C.POINTER(ScsSolution), C.POINTER(ScsInfo), c_int]
        L.oscs_update.restype = c_int
        L.oscs_update.argtypes = [C.c_void_p, PD, PD]
        L.oscs_finish.restype = None
        L.oscs_finish.argtypes = [C.c_void_p]
        L.oscs_set_default_settings.restype = None
        L.oscs_set_default_settings.argtypes = [C.POINTER(ScsSettings)]
        L.oscs_proj_cone.restype = c_int
        L.oscs_proj_cone.argtypes = [PD, C.POINTER(ScsCone), c_int, c_int]
        L.oscs_normalize.restype = c_int
        L.oscs_normalize.argtypes = [C.POINTER(ScsMatrix), C.POINTER(ScsMatrix), PD, PD,
                                     C.POINTER(ScsCone), PD, PD, PD, PD, PD]
        L.oscs_kkt_solve.restype = c_int
        L.oscs_kkt_solve.argtypes = [C.POINTER(ScsMatrix), C.POINTER(ScsMatrix), PD, PD, c_int,
                                     c_dbl, PI]
        L.oscs_spmv.restype = None
        L.oscs_spmv.argtypes = [C.POINTER(ScsMatrix), PD, PD, c_int]
        L.o_lin_sys_symbolic.restype = C.c_long
        L.o_lin_sys_symbolic.argtypes = [C.POINTER(ScsMatrix), C.POINTER(ScsMatrix), C.POINTER(C.c_long)]
        L.o_aa_init.restype = C.c_void_p
        L.o_aa_init.argtypes = [c_int, c_int, c_int, c_dbl, c_dbl, c_dbl, c_dbl]
        L.o_aa_apply.restype = c_dbl
        L.o_aa_apply.argtypes = [PD, PD, C.c_void_p]
        L.o_aa_safeguard.restype = c_int
        L.o_aa_safeguard.argtypes = [PD, PD, C.c_void_p]
        L.o_aa_reset.restype = None
        L.o_aa_reset.argtypes = [C.c_void_p]
        L.o_aa_free.restype = None
        L.o_aa_free.argtypes = [C.c_void_p]
        L.o_aa_get_stats.restype = None
        L.o_aa_get_stats.argtypes = [C.c_void_p, C.POINTER(ScsAaStats)]
        L.o_aa_last_gamma.restype = c_int
        L.o_aa_last_gamma.argtypes = [C.c_void_p, PD]
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _pd(a):
    return a.ctypes.data_as(PD)


def _pi(a):
    return a.ctypes.data_as(PI)


class _Keep(object):
    """holds numpy buffers alive next to the ctypes struct that points into them"""


def make_matrix(data, indices, indptr, m, n):
    k = _Keep()
    k.x, k.i, k.p = _f64(data), _i32(indices), _i32(indptr)
    k.mat = ScsMatrix(_pd(k.x), _pi(k.i), _pi(k.p), int(m), int(n))
    return k


def make_cone(cone):
    k = _Keep()
    z = int(cone.get("z", 0)) + int(cone.get("f", 0))
    k.bu = _f64(np.atleast_1d(cone.get("bu", [])))
    k.bl = _f64(np.atleast_1d(cone.get("bl", [])))
    if k.bu.size != k.bl.size:
        raise ValueError("bu different dimension to bl")
    k.q = _i32(np.atleast_1d(cone.get("q", [])))
    k.s = _i32(np.atleast_1d(cone.get("s", [])))
    k.cs = _i32(np.atleast_1d(cone.get("cs", [])))
    k.p = _f64(np.atleast_1d(cone.get("p", [])))
    bsize = k.bu.size + 1 if k.bu.size > 0 else 0
    k.cone = ScsCone(z, int(cone.get("l", 0)), _pd(k.bu), _pd(k.bl), bsize,
                     _pi(k.q), k.q.size, _pi(k.s), k.s.size, _pi(k.cs), k.cs.size,
                     int(cone.get("ep", 0)), int(cone.get("ed", 0)), _pd(k.p), k.p.size)
    return k


def cone_dims(cone):
    m = int(cone.get("z", 0)) + int(cone.get("f", 0)) + int(cone.get("l", 0))
    nb = len(np.atleast_1d(cone.get("bu", [])))
    m += nb + 1 if nb else 0
    m += int(np.sum(np.atleast_1d(cone.get("q", [])))) if len(np.atleast_1d(cone.get("q", []))) else 0
    for s in np.atleast_1d(cone.get("s", [])):
        m += int(s) * (int(s) + 1) // 2
    m += 3 * (int(cone.get("ep", 0)) + int(cone.get("ed", 0)) + len(np.atleast_1d(cone.get("p", []))))
    return m


_INT_KEYS = ("max_iters", "acceleration_lookback", "acceleration_interval", "acceleration_type_1")
_BOOL_KEYS = ("verbose", "normalize", "adaptive_scale")
_FLT_KEYS = ("scale", "eps_abs", "eps_rel", "eps_infeas", "alpha", "rho_x", "time_limit_secs",
             "acceleration_regularization", "acceleration_relaxation")


def make_settings(**settings):
    st = ScsSettings()
    lib().oscs_set_default_settings(C.byref(st))
    for k, v in settings.items():
        if k in _INT_KEYS:
            setattr(st, k, int(v))
        elif k in _BOOL_KEYS:
            setattr(st, k, 1 if v else 0)
        elif k in _FLT_KEYS:
            setattr(st, k, float(v))
        elif k in ("write_data_filename", "log_csv_filename", "linear_solver"):
            pass
        else:
            raise TypeError("unknown setting %r" % k)
    return st


def info_to_dict(info):
    d = {}
    for name, _ in ScsInfo._fields_:
        v = getattr(info, name)
        if name in ("status", "lin_sys_solver"):
            v = v.decode()
        elif name == "aa_stats":
            v = {n: getattr(info.aa_stats, n) for n, _ in ScsAaStats._fields_}
        d[name] = v
    return d


class OracleSCS(object):
    """CPU restatement with the raw-extension call surface (R:scs/scsobject.h:467-495)."""

    def __init__(self, shape, Ax, Ai, Ap, Px, Pi, Pp, b, c, cone, indirect=False, **settings):
        m, n = int(shape[0]), int(shape[1])
        self.m, self.n = m, n
        self._A = make_matrix(Ax, Ai, Ap, m, n)
        self._P = make_matrix(Px, Pi, Pp, n, n) if Px is not None else None
        self._b, self._c = _f64(b), _f64(c)
        self._k = make_cone(cone)
        st = make_settings(**settings)
        d = ScsData(m, n, C.pointer(self._A.mat), C.pointer(self._P.mat) if self._P else None,
                    _pd(self._b), _pd(self._c))
        self._work = lib().oscs_init(C.byref(d), C.byref(self._k.cone), C.byref(st), 1 if indirect else 0)
        if not self._work:
            raise ValueError("ScsWork allocation error!")
        self._x = np.zeros(n)
        self._y = np.zeros(m)
        self._s = np.zeros(m)

    def solve(self, warm_start=True, x=None, y=None, s=None):
        if warm_start:
            if x is not None:
                self._x[:] = x
            if y is not None:
                self._y[:] = y
            if s is not None:
                self._s[:] = s
        sol = ScsSolution(_pd(self._x), _pd(self._y), _pd(self._s))
        info = ScsInfo()
        lib().oscs_solve(self._work, C.byref(sol), C.byref(info), 1 if warm_start else 0)
        return {"x": self._x.copy(), "y": self._y.copy(), "s": self._s.copy(), "info": info_to_dict(info)}

    def update(self, b=None, c=None):
        bb = _f64(b) if b is not None else None
        cc = _f64(c) if c is not None else None
        lib().oscs_update(self._work, _pd(bb) if bb is not None else None, _pd(cc) if cc is not None else None)

    def __del__(self):
        w = getattr(self, "_work", None)
        if w:
            lib().oscs_finish(w)
            self._work = None


def solve(data, cone, indirect=False, warm_start=False, **settings):
    """Convenience: data dict {P?,A,b,c} with scipy CSC matrices."""
    from scipy import sparse
    A = sparse.csc_matrix(data["A"])
    A.sort_indices()
    P = data.get("P", None)
    Px = Pi = Pp = None
    if P is not None:
        P = sparse.triu(sparse.csc_matrix(P), format="csc")
        P.sort_indices()
        Px, Pi, Pp = P.data, P.indices, P.indptr
    s = OracleSCS(A.shape, A.data, A.indices, A.indptr, Px, Pi, Pp, data["b"], data["c"], cone,
                  indirect=indirect, **settings)
    return s.solve(warm_start=warm_start, x=data.get("x"), y=data.get("y"), s=data.get("s"))


def proj_cone(z, cone, dual=False):
    x = _f64(z).copy()
    k = make_cone(cone)
    rc = lib().oscs_proj_cone(_pd(x), C.byref(k.cone), x.size, 1 if dual else 0)
    if rc != 0:
        raise ValueError("bad cone")
    return x


def normalize(A, P, b, c, cone):
    """Returns (A_data_hat, P_data_hat, b_hat, c_hat, D, E, sigma, bl_hat, bu_hat)."""
    m, n = A.shape
    Am = make_matrix(A.data.copy(), A.indices, A.indptr, m, n)
    Pm = make_matrix(P.data.copy(), P.indices, P.indptr, n, n) if P is not None else None
    bb, cc = _f64(b).copy(), _f64(c).copy()
    k = make_cone(cone)
    D, E, sig = np.zeros(m), np.zeros(n), np.zeros(1)
    nb = max(k.bu.size, 1)
    bl, bu = np.zeros(nb), np.zeros(nb)
    lib().oscs_normalize(C.byref(Am.mat), C.byref(Pm.mat) if Pm else None, _pd(bb), _pd(cc),
                         C.byref(k.cone), _pd(D), _pd(E), _pd(sig), _pd(bl), _pd(bu))
    return Am.x, (Pm.x if Pm else None), bb, cc, D, E, float(sig[0]), bl[:k.bu.size], bu[:k.bu.size]


def kkt_solve(A, P, diag_r, rhs, indirect=False, tol=1e-12):
    m, n = A.shape
    Am = make_matrix(A.data, A.indices, A.indptr, m, n)
    Pm = make_matrix(P.data, P.indices, P.indptr, n, n) if P is not None else None
    r = _f64(rhs).copy()
    dr = _f64(diag_r)
    its = c_int(0)
    rc = lib().oscs_kkt_solve(C.byref(Am.mat), C.byref(Pm.mat) if Pm else None, _pd(dr), _pd(r),
                              1 if indirect else 0, float(tol), C.byref(its))
    if rc != 0:
        raise ValueError("factorisation failed")
    return r, its.value


def ldl_symbolic(A, P=None):
    """(nnz(L), height of the elimination tree) of the LDL' factor of the KKT pattern [[P + I, A'], [A, -I]] under the oracle's
    fill-reducing ordering — the symbolic phase only, no values (oscs_linsys.c o_lin_sys_symbolic)"""
    m, n = A.shape
    Am = make_matrix(A.data, A.indices, A.indptr, m, n)
    Pm = make_matrix(P.data, P.indices, P.indptr, n, n) if P is not None else None
    h = C.c_long(0)
    lnz = lib().o_lin_sys_symbolic(C.byref(Am.mat), C.byref(Pm.mat) if Pm else None, C.byref(h))
    return int(lnz), int(h.value)


def spmv(A, x, trans=False):
    m, n = A.shape
    Am = make_matrix(A.data, A.indices, A.indptr, m, n)
    xx = _f64(x)
    y = np.zeros(n if trans else m)
    lib().oscs_spmv(C.byref(Am.mat), _pd(xx), _pd(y), 1 if trans else 0)
    return y


class OracleAA(object):
    """oracle/oscs_aa.c driven step by step (the checker of tests/test_aa_gpu.py)."""
    _FIELDS = ("iter", "n_accept", "n_reject_lapack", "n_reject_rank0", "n_reject_nonfinite",
               "n_reject_weight_cap", "n_safeguard_reject", "last_rank", "last_aa_norm", "last_regularization")

    def __init__(self, dim, mem, type1=True, regularization=1e-8, relaxation=1.0, safeguard_factor=1.0,
                 max_weight_norm=1e10):
        self._h = lib().o_aa_init(int(dim), int(mem), 1 if type1 else 0, float(regularization), float(relaxation),
                                  float(safeguard_factor), float(max_weight_norm))
        self.dim = int(dim)

    def apply(self, f, x):
        ff = np.array(f, dtype=np.float64, copy=True)
        xx = _f64(x)
        nrm = lib().o_aa_apply(_pd(ff), _pd(xx), self._h)
        return nrm, ff

    def safeguard(self, f_new, x_new):
        ff = np.array(f_new, dtype=np.float64, copy=True)
        xx = np.array(x_new, dtype=np.float64, copy=True)
        rc = lib().o_aa_safeguard(_pd(ff), _pd(xx), self._h)
        return rc, ff, xx

    def reset(self):
        lib().o_aa_reset(self._h)

    def stats(self):
        st = ScsAaStats()
        lib().o_aa_get_stats(self._h, C.byref(st))
        return {k: getattr(st, k) for k in self._FIELDS}

    def last_gamma(self):
        g = np.zeros(64)
        n = lib().o_aa_last_gamma(self._h, _pd(g))
        return g[:n]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().o_aa_free(h)
