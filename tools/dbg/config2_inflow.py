"""config 2 (launch-bound: 15 us kernels) is 2.5 k iters/s alone but sometimes 0.9 k inside the default bench flow.
Reproducer: big solves first (as the bench's main line), then config 2 several times in the same process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
if not os.environ.get('NOTORCH'):
    import torch
import numpy as np
import scs
from scs import _scs_hip
import problem_gen as pg
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
common = dict(linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False, acceleration_lookback=10)
def run(workload, iters, reps, keep=None):
    K, n, k, seed = pg.workload(workload)
    data, _, _ = pg.gen_feasible(K, n, k, seed, proj, pattern=pg.workload_pattern(workload))
    out = []
    if SETTLE > 0:
        time.sleep(SETTLE)
    for r in range(reps):
        w = scs.SCS(data, K, max_iters=10, **common); w.solve()
        s = scs.SCS(data, K, max_iters=iters, **common)
        t = time.perf_counter(); sol = s.solve(); el = time.perf_counter() - t
        out.append(round(iters / el))
        if keep is not None: keep.append((w, s))
        del w, s
    print(workload, out, flush=True)
pre = sys.argv[1] if len(sys.argv) > 1 else "big"
SETTLE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
run("config2_lp_soc", 100, 3)
if pre == "short":  # for a kernel trace: few launches
    for _ in range(int(os.environ.get("CYCLES", "4"))):
        run("target_lp_soc", 25, 1)
        run("config2_lp_soc", 100, 2)
if pre == "long":  # is the slow mode tied to the solve or to the time since the big release?
    for _ in range(5):
        run("target_lp_soc", 25, 1)
        run("config2_lp_soc", 1000, 2)
if pre == "big":
    run("target_lp_soc", 25, 2)
    run("config2_lp_soc", 100, 4)
    run("config3_mixed", 5, 1)
    run("config2_lp_soc", 100, 4)
    run("config4_psd", 30, 1)
    run("config2_lp_soc", 100, 4)
    for _ in range(4):
        run("target_lp_soc", 25, 1)
        run("config2_lp_soc", 100, 2)
