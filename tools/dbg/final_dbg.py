import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scs-python_amd")):
    sys.path.insert(0, p)
import torch
import numpy as np
import scs
from scs import _scs_hip
import problem_gen as pg
torch.cuda.set_device(0)
K, n, k, seed = pg.workload("target_lp_soc")
data, p_star, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))
common = dict(eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False, acceleration_lookback=10, linear_solver="hip_indirect")
w = scs.SCS(data, K, max_iters=5, **common); w.solve(); del w
for rep in range(4):
    solver = scs.SCS(data, K, max_iters=20, **common)
    if rep % 2 == 0: solver._solver._set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sol = solver.solve(warm_start=False); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    i = sol["info"]
    print("profiling", rep % 2 == 0, "solve() wall %.1f ms, inside %.1f, sync after %.1f ms" % ((t1 - t0) * 1e3, i["solve_time"], (t2 - t1) * 1e3), flush=True)
    # wrapper pieces
    t = time.perf_counter(); a = sol["x"].copy(); b = sol["y"].copy(); c = sol["s"].copy(); print("  3 numpy copies %.1f ms" % ((time.perf_counter() - t) * 1e3))
