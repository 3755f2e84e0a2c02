"""`scs` — drop-in Python front end for the MI355X-native SCS hot path.

Mirrors the reference's pure-Python layer R:scs/py/__init__.py (L0/L1 of
SURVEY.md §1): `SCS(data, cone, **settings)` (:87-184), `.solve(warm_start, x,
y, s)` (:186-203), `.update(b, c)` (:205-214), legacy `solve(data, cone,
**settings)` (:218-230), the `LinearSolver` enum and its enum->module dispatch
(:28-74) and the status integers (:16-25).  Behaviour kept identical: argument
checks and their messages, CSC coercion with a warning, never mutating the
caller's matrices, upper-triangular extraction of P, `linear_solver` popped
before the backend sees the settings.

What differs: the backend modules shipped here are `scs._scs_hip`
(`LinearSolver.HIP_INDIRECT`) and `scs._scs_hip_dense` (`LinearSolver.HIP_DENSE`);
`AUTO` resolves — as the reference's does (R:scs/py/__init__.py:45-54: "the best
available direct solver") — to the direct one when it can take the problem and to
the indirect one otherwise (`_resolve_auto`); the CPU/CUDA members raise ImportError
exactly like an un-built optional backend of the reference does
(R:test/test_solve_random_cone_prob.py:24-30).
"""
import enum
import warnings
from importlib import import_module

import numpy as np
from scipy import sparse

from scs import _scs_hip

__version__ = _scs_hip.version()
__sizeof_int__ = _scs_hip.sizeof_int()
__sizeof_float__ = _scs_hip.sizeof_float()

# exit flags of scs_solve (R:scs/py/__init__.py:16-25)
INFEASIBLE_INACCURATE = -7
UNBOUNDED_INACCURATE = -6
SIGINT = -5
FAILED = -4
INDETERMINATE = -3
INFEASIBLE = -2
UNBOUNDED = -1
UNFINISHED = 0
SOLVED = 1
SOLVED_INACCURATE = 2


class LinearSolver(enum.Enum):
  """Which linear-system backend module `SCS` instantiates."""
  AUTO = "auto"
  QDLDL = "qdldl"
  CPU_INDIRECT = "cpu_indirect"
  MKL = "mkl"
  ACCELERATE = "accelerate"
  CPU_DENSE = "cpu_dense"
  GPU_INDIRECT = "gpu_indirect"
  CUDSS = "cudss"
  HIP_INDIRECT = "hip_indirect"  # MI355X (gfx950): device-resident indirect solver
  HIP_DENSE = "hip_dense"  # MI355X (gfx950): device dense direct solver for small problems (n <= 8192)


# enum member -> extension-module name under the `scs` package
_BACKEND_MODULES = {
    LinearSolver.QDLDL: "_scs_direct",
    LinearSolver.CPU_INDIRECT: "_scs_indirect",
    LinearSolver.MKL: "_scs_mkl",
    LinearSolver.ACCELERATE: "_scs_accelerate",
    LinearSolver.CPU_DENSE: "_scs_dense",
    LinearSolver.GPU_INDIRECT: "_scs_gpu",
    LinearSolver.CUDSS: "_scs_cudss",
    LinearSolver.HIP_INDIRECT: "_scs_hip",
    LinearSolver.HIP_DENSE: "_scs_hip_dense",
}


def _load_module(name):
  return import_module("scs." + name)


# AUTO's view of the dense direct solver (csrc/dense.hpp): the order up to which it wins a whole solve against the indirect path
# (measured: profiles/r06_auto_crossover.txt — 2.4 x at n = 1350, 1.3 x at 4096, 0.6-0.7 x at 6144 / 8192, where the 2 n^3-flop
# re-inversions at every adaptive-scale update and the 8 n^2-byte product per iteration cost more than ~10 launch-bound PCG steps;
# `LinearSolver.HIP_DENSE` by name accepts n <= 8192), the share of the free HBM its inverse may take and the work of forming
# G = R_x + P + A' R_y^-1 A (one product per pair of nonzeros of a row of A) it is worth paying at every scale update.
_AUTO_DENSE_MAX_N = 4096
_AUTO_DENSE_HBM_SHARE = 0.25
_AUTO_DENSE_MAX_BUILD_PRODUCTS = 2e9


def _dense_direct_fits(m, n, A):
  """True when the device's direct solver can take an m x n problem with the CSC matrix A."""
  if n > _AUTO_DENSE_MAX_N:
    return False
  info = _scs_hip.mem_info()
  if info is None:  # no device: let the indirect module report it ("ScsWork allocation error! (no HIP device ...)")
    return False
  npad = -(-n // 64) * 64
  if 8.0 * npad * npad > _AUTO_DENSE_HBM_SHARE * info[0]:
    return False
  if A is not None and A.nnz:
    try:
      row_len = np.bincount(A.indices, minlength=m).astype(np.float64)
    except ValueError:  # negative row indices: the backend's own validation reports them
      return False
    if float(row_len @ row_len) > _AUTO_DENSE_MAX_BUILD_PRODUCTS:
      return False
  return True


def _resolve_auto(m=None, n=None, A=None):
  """AUTO = the best DIRECT solver that is usable, as in the reference (R:scs/py/__init__.py:45-54: MKL Pardiso, else the
  bundled QDLDL).  Here: the dense direct solver of the device when the problem fits it (n <= 4096, the n x n inverse within a
  quarter of the free HBM, G cheap to form), else the indirect solver — a sparse factorisation of the BASELINE patterns fills to
  0.06 N^2 and does not exist on the device."""
  if n is not None and _dense_direct_fits(m, n, A):
    return _load_module("_scs_hip_dense")
  return _scs_hip


def _select_scs_module(stgs, m=None, n=None, A=None):
  """Pop `linear_solver` (enum member or its string value) and load that backend."""
  choice = stgs.pop("linear_solver", LinearSolver.AUTO)
  if isinstance(choice, str):
    choice = LinearSolver(choice)
  if choice is LinearSolver.AUTO:
    return _resolve_auto(m, n, A)
  if choice is LinearSolver.HIP_INDIRECT:
    return _scs_hip
  return _load_module(_BACKEND_MODULES[choice])


def _has_lower_tri(P):
  """True when a sorted CSC matrix stores anything strictly below the diagonal."""
  counts = np.diff(P.indptr)
  cols = np.flatnonzero(counts)
  if cols.size == 0:
    return False
  bottom = P.indices[P.indptr[cols + 1] - 1]  # largest row index of each non-empty column
  return bool((bottom > cols).any())


def _csc_sorted(M, what):
  """CSC with sorted indices, never touching the caller's object."""
  if M.format != "csc":
    warnings.warn("Converting %s to a CSC (compressed sparse column) matrix; may take a while." % what)
    M = M.tocsc()
  if not M.has_sorted_indices:
    M = M.sorted_indices()  # a copy; sort_indices() would mutate the caller's matrix
  return M


def _dense_1d(v):
  if sparse.issparse(v):
    return np.asarray(v.todense()).ravel()
  return v


class SCS(object):

  def __init__(self, data, cone, **settings):
    """Set up a solver workspace.

    @param data     dict with `A`, `b`, `c` and optionally `P`.
    @param cone     dict describing the cone K.
    @param settings solver settings as keyword arguments, plus `linear_solver`.
    """
    self._settings = settings
    if not data or not cone:
      raise ValueError("Missing data or cone information")
    if "b" not in data or "c" not in data:
      raise ValueError("Missing one of b, c from data dictionary")
    if "A" not in data:
      raise ValueError("Missing A from data dictionary")
    A, b, c = data["A"], data["b"], data["c"]
    if A is None or b is None or c is None:
      raise ValueError("Incomplete data specification")
    if not sparse.issparse(A):
      raise TypeError("A is required to be a sparse matrix")
    A = _csc_sorted(A, "A")
    b, c = _dense_1d(b), _dense_1d(c)
    m, n = len(b), len(c)
    if A.shape != (m, n):
      raise ValueError("A shape not compatible with b,c")

    Px = Pi = Pp = None
    P = data.get("P", None)
    if P is not None:
      if not sparse.issparse(P):
        raise TypeError("P is required to be a sparse matrix")
      if P.shape != (n, n):
        raise ValueError("P shape not compatible with A,b,c")
      P = _csc_sorted(P, "P")
      if _has_lower_tri(P):  # the core wants the upper triangle only
        P = sparse.triu(P, format="csc")
      Px, Pi, Pp = P.data, P.indices, P.indptr

    backend = _select_scs_module(self._settings, m, n, A)
    self._solver = backend.SCS((m, n), A.data, A.indices, A.indptr, Px, Pi, Pp, b, c, cone,
                               **self._settings)

  def solve(self, warm_start=True, x=None, y=None, s=None):
    """Run the solver.

    @param warm_start  start from the previous solution (or from x, y, s when given).
    @return dict with keys 'x', 'y', 's', 'info'.
    """
    return self._solver.solve(warm_start, x, y, s)

  def update(self, b=None, c=None):
    """Replace `b` and/or `c`, re-using the workspace for the next solve."""
    self._solver.update(b, c)


def solve(data, cone, **settings):
  """Legacy one-shot API; warm-start vectors may ride along in `data`."""
  solver = SCS(data, cone, **settings)
  return solver.solve(warm_start=True, x=data.get("x"), y=data.get("y"), s=data.get("s"))


def solve_batch(solvers, warm_start=False):
  """Solve several `SCS` objects of the HIP backend as one batch: equally shaped problems advance in lock step and
  share every kernel launch (the MI355X answer to the reference's "independent instances run concurrently",
  R:test/test_thread_safety.py:78-93).  Returns one result dict per solver, identical to what `.solve()` returns."""
  raw = []
  for sv in solvers:
    if not isinstance(sv, SCS) or not isinstance(sv._solver, _scs_hip.SCS):
      raise TypeError("solve_batch needs scs.SCS objects created with LinearSolver.HIP_INDIRECT")
    raw.append(sv._solver)
  return _scs_hip.solve_batch(raw, warm_start)
