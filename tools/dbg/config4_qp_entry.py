"""config-4-shaped QP (tests/test_hip_fullsize.py) under the fixed and the residual-tied PSD stopping level: iterations,
status, entry-wise error of (x, y, s) against the constructed solution.  usage: python tools/dbg/config4_qp_entry.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scs-python_amd"))
import problem_gen as pg, helpers, scs
K = {"l": 1000, "s": [200] * 50}
data, p_star, (x0, y0, s0) = pg.gen_feasible_qp(K, 335000, 30, 44, helpers.proj_dual_l_s_numpy)
for eps in (1e-8, 1e-9):
    t = time.time()
    sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=eps, eps_rel=eps, verbose=False, max_iters=20000).solve()
    i = sol["info"]
    errs = {k: float(np.abs(sol[k] - r).max() / np.abs(r).max()) for k, r in (("x", x0), ("y", y0), ("s", s0))}
    print(os.environ.get("SCS_HIP_PSD_TOL", "adaptive"), os.environ.get("SCS_HIP_PSD_TOL_K", "dflt"), eps, i["status"], i["iter"],
          "pobj err %.2e" % (abs(i["pobj"] - p_star) / max(1, abs(p_star))), errs, "%.1fs" % (time.time() - t), flush=True)
