"""config-5 per-problem workload: HIP (grouped) vs the oracle's LDL' at several tolerances (debugging aid)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import scs
from scs import _scs_hip
from oracle import scs_oracle
import problem_gen as pg

proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
Kb, nb, kb, seed = pg.workload("config5_small")
N = int(os.environ.get("NPROB", "4"))
gen = [pg.gen_feasible(Kb, nb, kb, seed + i, proj) for i in range(N)]
for eps in (1e-4, 1e-6, 1e-8):
    stg = dict(verbose=False, eps_abs=eps, eps_rel=eps, max_iters=int(os.environ.get("MAXIT", "40000")))
    t = time.time()
    grp = scs.solve_batch([scs.SCS(g[0], Kb, **stg) for g in gen])
    tg = time.time() - t
    for i, (g, b) in enumerate(zip(gen, grp)):
        t = time.time()
        ref = scs_oracle.solve(g[0], Kb, indirect=False, **stg)
        to = time.time() - t
        xs, ys, ss = g[2]
        def rel(a, r): return np.linalg.norm(a - r) / np.linalg.norm(r), np.abs(a - r).max() / np.abs(r).max()
        print("eps %.0e seed %d: hip iters %d (%s) oracle iters %d (%s, %.1fs) | hip-vs-oracle x %.1e/%.1e y %.1e/%.1e s %.1e/%.1e | hip-vs-constructed y %.1e/%.1e | oracle-vs-constructed y %.1e/%.1e"
              % (eps, seed + i, b["info"]["iter"], b["info"]["status"], ref["info"]["iter"], ref["info"]["status"], to,
                 *rel(b["x"], ref["x"]), *rel(b["y"], ref["y"]), *rel(b["s"], ref["s"]), *rel(b["y"], ys), *rel(ref["y"], ys)))
    print("group wall %.2fs" % tg)
