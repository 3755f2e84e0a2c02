"""Why does BASELINE config 3 need hundreds of CG steps per ADMM iteration?  (bench hygiene)
Same matrix, same 20 cold iterations; only the cone labels of the first 100,000 rows / the settings change.
    python tools/config3_cg_study.py > profiles/r03_config3_cg_study.txt      (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import numpy as np
import scs
from scs import _scs_hip
import problem_gen as pg

proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config3_mixed")
data, p_star, _ = pg.gen_feasible(K, n, k, seed, proj)
base = dict(eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False, acceleration_lookback=10, max_iters=20)


def run(tag, cone, **kw):
    stg = dict(base, **kw)
    sol = scs.SCS(data, cone, **stg).solve(warm_start=False)
    i = sol["info"]
    print("%-78s %7.1f CG steps / iteration, %7.2f ms / iteration" % (tag, i["cg_iters"] / i["iter"], i["solve_time"] / i["iter"]))


print("# config 3 (m = 999,999, n = 500,000, nnz = 1e7): CG steps of the first 20 ADMM iterations (cold start, eps = 0)")
print("# R_y = 1 / (1000 scale) on zero-cone rows and 1 / scale elsewhere (SURVEY App. A): with scale = 0.1 the reduced system")
print("# rho_x I + A' R_y^-1 A weighs the 100,000 equality rows 1000 x heavier than the 900,000 others -> condition number")
print("# ~ 1e3 (rank-1e5 part on top of a well-conditioned one), Jacobi-preconditioned CG needs ~ sqrt(kappa) log(1/tol) steps.")
run("config 3 as specified (z = 100,000)", K)
Kl = dict(K)
Kl["l"] = K["l"] + K["z"]
Kl["z"] = 0
run("same A, b, c; the 100,000 z rows declared `l` (no 1000 x weight; a different problem)", Kl)
run("config 3, scale = 1e-3 (R_y^-1 = 1 on z rows, 1e-3 elsewhere: same 1000 x ratio)", K, scale=1e-3, adaptive_scale=False)
run("config 3, rho_x = 1e-2 (larger ridge on the x block)", K, rho_x=1e-2)
Kn = {kk: vv for kk, vv in K.items() if kk not in ("ep", "ed", "p")}
Kn["l"] = K["l"] + 3 * (K["ep"] + K["ed"] + len(K["p"]))
run("same A; exp / pow rows declared `l` (z kept): the nonlinear cones are not the cause", Kn)
