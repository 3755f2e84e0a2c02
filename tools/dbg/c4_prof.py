"""config 4, one solve stopped at iteration MAXIT (default 225): the workload for a kernel trace of the window past the cold start"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import scs, problem_gen as pg
from scs import _scs_hip
K, n, k, seed = pg.workload("config4_psd")
d = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))[0]
s = scs.SCS(d, K, verbose=False, eps_abs=0., eps_rel=0., eps_infeas=0., max_iters=int(os.environ.get("MAXIT", "225")))
r = s.solve()
st = s._solver._psd_refine_stats()
print("iters %d solve %.1f ms refined/matrix %.1f failed %.2f" % (r["info"]["iter"], r["info"]["solve_time"], st[:, 0].mean(), st[:, 1].mean()))
