"""first ADMM iteration at which a grouped solve differs from separate solves (debugging aid)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import scs
from scs import _scs_hip
import problem_gen as pg

proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K = {"l": 300, "q": [12] * 6, "s": [6] * 4}
probs = [pg.gen_feasible(K, 120, 12, 4100 + i, proj)[0] for i in range(int(os.environ.get("NPROB", "3")))]
extra = eval(os.environ.get("STG", "{}"))
for k in [1, 2, 5, 9, 10, 11, 12, 19, 20, 21, 22, 25, 26, 27, 30, 31, 40, 41, 50, 51, 75, 76, 100, 101, 102, 110, 111, 125, 126, 150, 200, 300]:
    stg = dict(verbose=False, max_iters=k, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, **extra)
    solo = [scs.SCS(d, K, **stg).solve(warm_start=False) for d in probs]
    grp = scs.solve_batch([scs.SCS(d, K, **stg) for d in probs])
    bad = [i for i, (a, b) in enumerate(zip(solo, grp)) if not (np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]) and np.array_equal(a["s"], b["s"]))]
    print(k, "differs:" if bad else "same", bad, [(a["info"]["cg_iters"], b["info"]["cg_iters"]) for a, b in zip(solo, grp)],
          [(a["info"]["aa_stats"]["n_accept"], b["info"]["aa_stats"]["n_accept"], a["info"]["scale_updates"], b["info"]["scale_updates"]) for a, b in zip(solo, grp)])
    if bad and not os.environ.get("ALL"):
        break
