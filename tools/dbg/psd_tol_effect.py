"""does the residual-tied stopping level of the PSD sweeps change the ADMM iteration count?  (run twice: SCS_HIP_PSD_TOL=fixed / unset)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import scs
from scs import _scs_hip
import problem_gen as pg
import helpers
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
mode = os.environ.get("SCS_HIP_PSD_TOL", "adaptive")
for fname, prefix in (("problems_sdp.npz", "feas0_"), ("problems_sdp.npz", "feas1_"), ("problems_sdp.npz", "feas2_"), ("problems_std.npz", "std_feas_")):
    data, K, p_star = helpers.load_problem(fname, prefix)
    for eps in (1e-4, 1e-9):
        t = time.time()
        sol = scs.SCS(data, K, verbose=False, eps_abs=eps, eps_rel=eps).solve()
        print("%-9s %s%-10s eps %.0e: %-8s iters %6d  pobj err %.2e  %.2fs" % (mode, fname[9:12], prefix, eps, sol["info"]["status"], sol["info"]["iter"],
              abs(sol["info"]["pobj"] - p_star) / max(1, abs(p_star)), time.time() - t))
K, n, k, seed = pg.workload("config4_psd")
data, p_star, _ = pg.gen_feasible(K, n, k, seed, proj)
for eps in (1e-4, 1e-6):
    t = time.time()
    sol = scs.SCS(data, K, verbose=False, eps_abs=eps, eps_rel=eps).solve()
    print("%-9s config4 eps %.0e: %-8s iters %6d  pobj err %.2e  solve %.2fs" % (mode, eps, sol["info"]["status"], sol["info"]["iter"],
          abs(sol["info"]["pobj"] - p_star) / max(1, abs(p_star)), sol["info"]["solve_time"] * 1e-3))
