"""full solve of the power-law workload: status, iterations, times"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import scs, problem_gen as pg
from scs import _scs_hip as hip
t = time.time()
K, n, k, seed = pg.workload("powerlaw_lp")
data, p_star, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: hip.proj_cone(z, K, dual=True), pattern=pg.workload_pattern("powerlaw_lp"))
print("gen %.1f s" % (time.time() - t), flush=True)
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=eps, eps_rel=eps, verbose=False, max_iters=int(sys.argv[2]) if len(sys.argv) > 2 else 3000).solve()
i = sol["info"]
print(i["status"], i["iter"], "solve %.1f s" % (i["solve_time"] / 1e3), "cg/iter %.1f" % (i["cg_iters"] / max(i["iter"], 1)), "pobj rel err %.2e" % (abs(i["pobj"] - p_star) / abs(p_star)),
      "res", i["res_pri"], i["res_dual"], i["gap"], i["lin_sys_solver"])
