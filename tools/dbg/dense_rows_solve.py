"""the LP with a budget row and a dense column of tests/test_hip_parity.py::test_solve_with_dense_rows_stays_on_the_column_sorted_layout:
solve time / iterations / SpMV time per layout policy"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import problem_gen as pg, helpers
from scs import _scs_hip as hip
proj = lambda z, K: hip.proj_cone(z, K, dual=True)
K = {"l": 120000, "q": [10] * 2000}
data, p_star, _ = pg.gen_feasible(K, 70000, 16, 5, proj)
rng = np.random.default_rng(3)
A = data["A"].tolil()
A[5, :] = 0.05 * rng.standard_normal(A.shape[1])
A[:, 9] = 0.05 * rng.standard_normal(A.shape[0]).reshape(-1, 1)
A = A.tocsc(); A.sort_indices()
x0 = rng.standard_normal(A.shape[1]); z = rng.standard_normal(A.shape[0])
y0 = proj(z, K); s0 = y0 - z
dat = {"A": A, "b": A @ x0 + s0, "c": -(A.T @ y0)}
t = time.time()
sol = hip.SCS(*helpers.raw_args(dat, K), eps_abs=1e-7, eps_rel=1e-7, verbose=False).solve(False, None, None, None)
i = sol["info"]
print(os.environ.get("TAG", ""), i["lin_sys_solver"], i["status"], i["iter"], "solve %.2f s" % (i["solve_time"] / 1e3), "setup %.2f s" % (i["setup_time"] / 1e3),
      "cg", i["cg_iters"], "wall %.1f" % (time.time() - t), flush=True)
xs = rng.standard_normal(A.shape[1]); ys = rng.standard_normal(A.shape[0])
for tr, v in ((False, xs), (True, ys)):
    hip.spmv(A, v, transpose=tr)
    t = time.time()
    for _ in range(20): hip.spmv(A, v, transpose=tr)
    print("   spmv transpose=%s (incl. layout build + copies) %.1f ms per call" % (tr, (time.time() - t) / 20 * 1e3))
