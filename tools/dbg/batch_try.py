import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scs-python_amd")):
    sys.path.insert(0, p)
import numpy as np
import scs
from scs import _scs_hip, batch
import problem_gen as pg
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
Kb, nb_, kb_, seedb = pg.workload("config5_small")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
probs = []
for i in range(N):
    d, _, _ = pg.gen_feasible(Kb, nb_, kb_, seedb + i, proj)
    probs.append((d, Kb, dict(verbose=False)))
scs.SCS(probs[0][0], Kb, verbose=False, max_iters=50).solve()
for th in [int(t) for t in sys.argv[2].split(",")]:
    t = time.time(); res = batch.solve_sharded(probs, threads=th); dt = time.time() - t
    its = sum(r["info"]["iter"] for r in res); ok = sum(r["info"]["status_val"] == 1 for r in res)
    print(os.environ.get("SCS_HIP_PERSIST", "-"), "threads", th, "solved", ok, "iters", its, "wall %.2f s => %.0f ADMM iters/s" % (dt, its / dt), flush=True)
