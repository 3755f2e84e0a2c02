import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import helpers, problem_gen as pg
from scs import _scs_hip as hip
from oracle import scs_oracle as oracle
K, n, k = {"z": 120, "l": 200, "q": [12, 7, 30]}, 150, 8
data, p_star, _ = pg.gen_feasible(K, n, k, 31, lambda z, K: oracle.proj_cone(z, K, dual=True))
sol = hip.SCS(*helpers.raw_args(data, K), eps_abs=1e-9, eps_rel=1e-9, eps_infeas=1e-9, verbose=False, max_iters=int(os.environ.get("MAXIT", "40"))).solve(False, None, None, None)
print(sol["info"]["iter"], sol["info"]["status"], sol["info"]["cg_iters"] if "cg_iters" in sol["info"] else "")
