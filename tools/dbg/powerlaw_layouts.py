"""150 plain ADMM iterations of the power-law workload on whichever layout the environment selects: fingerprints of the iterates"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import scs, problem_gen as pg
from scs import _scs_hip as hip
K, n, k, seed = pg.workload("powerlaw_lp")
data, p_star, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: hip.proj_cone(z, K, dual=True), pattern=pg.workload_pattern("powerlaw_lp"))
its = int(sys.argv[1]) if len(sys.argv) > 1 else 150
sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, max_iters=its, acceleration_lookback=0,
              verbose=False).solve()
i = sol["info"]
print(os.environ.get("TAG", "?"), i["lin_sys_solver"].split("(")[1][:45], "cg", i["cg_iters"], "x[:3]", np.round(sol["x"][:3], 6), "|x|", round(float(np.linalg.norm(sol["x"])), 6),
      "pobj %.8g res_pri %.6g res_dual %.6g" % (i["pobj"], i["res_pri"], i["res_dual"]), flush=True)
