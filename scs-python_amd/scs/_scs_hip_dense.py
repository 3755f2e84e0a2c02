"""Backend module `scs._scs_hip_dense` — the MI355X DENSE DIRECT linear-system solver (csrc/dense.hpp) behind the same raw
`SCS` type as `scs._scs_hip`.  The reference selects its linear solver by extension module (R:scs/py/__init__.py:40-66:
`_scs_direct`, `_scs_indirect`, `_scs_dense`, `_scs_gpu`, `_scs_cudss` ...; each exports SCS, version, sizeof_int,
sizeof_float: R:scs/scsmodule.h:16-23); this is the module `LinearSolver.HIP_DENSE` loads.  For problems with n <= 8192:
the explicit inverse of the reduced KKT matrix R_x + P + A' R_y^-1 A lives in HBM, the linear solve of an ADMM iteration
is three dependent launches and plain iterations never wait for the host."""
from . import _scs_hip
from ._scs_hip import version, sizeof_int, sizeof_float, device_count, last_error  # noqa: F401


class SCS(_scs_hip.SCS):
    _LINSYS = 2


def solve_batch(solvers, warm_start=False):
    return _scs_hip.solve_batch(solvers, warm_start)
