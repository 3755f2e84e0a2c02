import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np, helpers, problem_gen as pg
from scs import _scs_hip as hip, _scs_hip_dense as dense
from oracle import scs_oracle as oracle
K = {"z": 6, "l": 80, "bu": [1.0, 2.0], "bl": [-1.0, -0.5], "q": [7, 9], "s": [5], "cs": [3], "ep": 3, "ed": 2, "p": [0.4, -0.7]}
data, p_star, _ = pg.gen_feasible_qp(K, 70, 6, 2718, lambda z, K: oracle.proj_cone(z, K, dual=True))
args = helpers.raw_args(data, K)
stg = dict(eps_abs=1e-8, eps_rel=1e-8, eps_infeas=1e-9, verbose=False, max_iters=50000, adaptive_scale=False)
prev = None
for rep in range(3):
    for name, mod in (("dense", dense), ("indirect", hip)):
        r = mod.SCS(*args, **stg).solve(False, None, None, None)
        print(rep, name, r["info"]["status"], r["info"]["iter"], "%.12f" % r["info"]["pobj"], "x0 %.17g" % r["x"][0], flush=True)
r = oracle.OracleSCS(*args, indirect=False, **stg).solve(False)
print("oracle ldl", r["info"]["status"], r["info"]["iter"], "%.12f" % r["info"]["pobj"])
r = oracle.OracleSCS(*args, indirect=True, **stg).solve(False)
print("oracle cg", r["info"]["status"], r["info"]["iter"], "%.12f" % r["info"]["pobj"])
