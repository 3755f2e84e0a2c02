import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import scs, problem_gen as pg, helpers
from oracle import scs_oracle as oracle
K = {"l": 300}
data, p_star, _ = pg.gen_feasible(K, 120, 8, 77, lambda z, K: oracle.proj_cone(z, K, dual=True))
data["b"] = data["b"] * 1e3
base = dict(eps_abs=1e-9, eps_rel=1e-9, eps_infeas=1e-9, verbose=False, scale=1e-3)
def show(tag, r):
    i = r["info"]
    print("%-34s %-40s iter %6d scale_updates %2d scale %.3e res_pri %.2e res_dual %.2e gap %.2e pobj %.9f" % (tag, i["status"], i["iter"], i["scale_updates"], i["scale"], i["res_pri"], i["res_dual"], i["gap"], i["pobj"]))
args = helpers.raw_args(data, K)
show("oracle LDL", oracle.OracleSCS(*args, indirect=False, **base).solve(False))
show("oracle CG", oracle.OracleSCS(*args, indirect=True, **base).solve(False))
for ls in ("hip_indirect", "hip_dense"):
    show(ls, scs.SCS(data, K, linear_solver=ls, **base).solve())
    show(ls + " no AA", scs.SCS(data, K, linear_solver=ls, acceleration_lookback=0, **base).solve())
    show(ls + " no adaptive scale", scs.SCS(data, K, linear_solver=ls, adaptive_scale=False, **base).solve())
show("oracle LDL no AA", oracle.OracleSCS(*args, indirect=False, acceleration_lookback=0, **base).solve(False))
