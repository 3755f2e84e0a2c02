"""12 plain ADMM iterations of a large workload on whichever layout the environment selects: fingerprints (run once per layout)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import scs, problem_gen as pg, helpers
from scs import _scs_hip as hip
name = sys.argv[1]
proj = lambda z, K: hip.proj_cone(z, K, dual=True)
if name == "lp2x":  # twice the metric workload
    K = {"l": 4000000}
    data, p_star, _ = pg.gen_feasible(K, 2000000, 20, 41, proj)
elif name == "qp":
    K = {"l": 600000, "q": [10] * 20000}
    data, p_star, _ = pg.gen_feasible_qp(K, 400000, 8, 17, proj)
else:
    K, n, k, seed = pg.workload(name)
    data, p_star, _ = pg.gen_feasible(K, n, k, seed, proj, pattern=pg.workload_pattern(name))
sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, max_iters=12, acceleration_lookback=0, verbose=False).solve()
i = sol["info"]
print("%-8s %-14s %-40s cg %4d |x| %.10g |y| %.10g pobj %.10g res_pri %.8g res_dual %.8g" % (os.environ.get("TAG", "?"), name, i["lin_sys_solver"].split("(")[1][:40], i["cg_iters"],
      np.linalg.norm(sol["x"]), np.linalg.norm(sol["y"]), i["pobj"], i["res_pri"], i["res_dual"]), flush=True)
