"""config 4: what K9's gate sees as the matrices arrive (|K1|_F, |off| / |A|, omega of S = V'AV before any sweep), by ADMM iteration"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import numpy as np
import scs, problem_gen as pg
from scs import _scs_hip
K, n, k, seed = pg.workload("config4_psd")
d = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))[0]
for it in (30, 60, 104, 108, 115, 125, 135, 150, 175, 200, 225, 250, 300, 400, 500):
    s = scs.SCS(d, K, verbose=False, eps_abs=0., eps_rel=0., eps_infeas=0., max_iters=it)
    r = s.solve()
    st = s._solver._psd_refine_stats()
    print("iteration %4d: |K1|_F median %.1e max %.1e   |off|/|A| median %.1e max %.1e   omega median %.1e max %.1e   refined so far %.0f" % (
        it, np.sqrt(np.median(st[:, 5])), np.sqrt(st[:, 5].max()), np.sqrt(np.median(st[:, 6])), np.sqrt(st[:, 6].max()), np.median(st[:, 7]), st[:, 7].max(), st[:, 0].mean()), flush=True)
