"""scs_init of one config-5 member: wall time of consecutive constructions (the first pays the process's first-use costs), and the phase
marks of SCS_HIP_DEBUG=setup for the last.  python tools/dbg/init_time.py [dense|indirect] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import scs, problem_gen as pg
from scs import _scs_hip
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config5_small")
mode = sys.argv[1] if len(sys.argv) > 1 else "dense"
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 6
LS = scs.LinearSolver.HIP_DENSE if mode == "dense" else scs.LinearSolver.HIP_INDIRECT
datas = [pg.gen_feasible(K, n, k, seed + i, proj)[0] for i in range(cnt)]
keep = []
for i, d in enumerate(datas):
    if i == cnt - 1:
        os.environ["SCS_HIP_DEBUG"] = "setup"
    t = time.perf_counter()
    s = scs.SCS(d, K, linear_solver=LS, verbose=False)
    el = time.perf_counter() - t
    keep.append(s)
    print("init %d: %.2f ms" % (i, el * 1e3), flush=True)
if not os.environ.get("INIT_ONLY"):
    t = time.perf_counter(); r = keep[0].solve(); print("first solve (incl. lazy dense setup): %.2f ms, %d iterations" % ((time.perf_counter() - t) * 1e3, r["info"]["iter"]))
