"""per-iteration latency of ONE config-5 problem (the straggler of the batch): run-ahead loop, hipGraph loop, and a grouped solve of 2"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import scs, problem_gen as pg
from scs import _scs_hip
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config5_small")
datas = [pg.gen_feasible(K, n, k, seed + i, proj)[0] for i in range(2)]
mode = sys.argv[1]
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
LS = scs.LinearSolver.HIP_DENSE if mode.startswith("dense") else scs.LinearSolver.HIP_INDIRECT
kw = dict(linear_solver=LS, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, max_iters=ITERS, verbose=False)
if mode in ("group2", "dense_group2"):
    ss = [scs.SCS(d, K, **kw) for d in datas]
    scs.solve_batch(ss)
    t = time.perf_counter(); r = scs.solve_batch(ss); el = time.perf_counter() - t
    print(mode, "%.1f us per lock-step iteration" % (el / ITERS * 1e6), r[0]["info"]["lin_sys_solver"])
else:
    s = scs.SCS(datas[0], K, **kw); s.solve()
    t = time.perf_counter(); r = s.solve(); el = time.perf_counter() - t
    print(mode, "%.1f us per iteration" % (el / ITERS * 1e6), r["info"]["lin_sys_solver"], "cg/iter %.2f" % (r["info"]["cg_iters"] / ITERS))
