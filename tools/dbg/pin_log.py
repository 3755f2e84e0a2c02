"""one config-2 init + 3 iterations under AMD_LOG_LEVEL=4: which runtime calls pin host memory?  (stderr is the log)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import scs
from scs import _scs_hip
import problem_gen as pg
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config2_lp_soc")
data, _, _ = pg.gen_feasible(K, n, k, seed, proj)
print("=== init", file=sys.stderr, flush=True)
s = scs.SCS(data, K, max_iters=3, linear_solver=scs.LinearSolver.HIP_INDIRECT, verbose=False)
print("=== solve", file=sys.stderr, flush=True)
s.solve()
print("=== del", file=sys.stderr, flush=True)
del s
print("=== end", file=sys.stderr, flush=True)
