"""a uniformly denser matrix (100 nonzeros per row of A', none clustered) must keep the column-sorted
layout with its rows IN the passes, not peeled into the side launch.  Prints the layout and kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import numpy as np
import scs
from scs import _scs_hip
import problem_gen as pg
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
m, n, k = 1000000, 200000, 100   # A' rows (= columns of A) have 100 nonzeros, A rows 20
K = {"l": m}
data, _, _ = pg.gen_feasible(K, n, k, 21, proj)
sv = scs.SCS(data, K, verbose=False, max_iters=30, eps_abs=0.0, eps_rel=0.0)
sv._solver._set_profiling(True)
sol = sv.solve(warm_start=False)
kt = sv._solver._time_matvec(reps=20)
print(sol["info"]["lin_sys_solver"], "| K1 %.1f us, K2 %.1f us | nnz %d" % (kt["k1_ms"] * 1e3, kt["k2_ms"] * 1e3, data["A"].nnz))
y = _scs_hip.spmv(data["A"], np.ones(n)); yt = _scs_hip.spmv(data["A"], np.ones(m), transpose=True)
print("spmv vs scipy:", np.abs(y - data["A"] @ np.ones(n)).max(), np.abs(yt - data["A"].T @ np.ones(m)).max())
