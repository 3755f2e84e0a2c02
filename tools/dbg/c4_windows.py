"""config 4: iterations/s of the windows [0,105), [105,225) and of a whole solve to 1e-4, plus the refinement statistics of K9 (round 5)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import numpy as np
import scs, problem_gen as pg
from scs import _scs_hip
K, n, k, seed = pg.workload("config4_psd")
d = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))[0]
tag = "REFINE=%s" % os.environ.get("SCS_HIP_PSD_REFINE", "1")
for rep in range(2):
    s = scs.SCS(d, K, verbose=False, eps_abs=0., eps_rel=0., eps_infeas=0., max_iters=225)
    s._solver._set_mark(105)
    r = s.solve()
    mk = s._solver._get_mark()
    i = r["info"]
    st = s._solver._psd_refine_stats()
    print("%s rep %d: [0,105) %.1f ms = %.0f it/s   [105,225) %.1f ms = %.0f it/s   refined calls/matrix %.1f failed %.1f" % (
        tag, rep, mk["ms"], 105e3 / mk["ms"], i["solve_time"] - mk["ms"], 120e3 / (i["solve_time"] - mk["ms"]), st[:, 0].mean(), st[:, 1].mean()), flush=True)
s = scs.SCS(d, K, verbose=False, max_iters=3000)
t0 = time.time()
r = s.solve()
t1 = time.time()
st = s._solver._psd_refine_stats()
i = r["info"]
print("%s whole solve: %s, %d iterations, %.3f s = %.0f it/s; pobj %.9f; refined calls/matrix %.1f (min %d max %d), failed %.2f, last mixed-off %.1e" % (
    tag, i["status"], i["iter"], i["solve_time"] / 1e3, i["iter"] / (i["solve_time"] / 1e3), i["pobj"], st[:, 0].mean(), st[:, 0].min(), st[:, 0].max(), st[:, 1].mean(),
    np.sqrt(st[:, 3].max())), flush=True)
tp = s._solver._time_psd(20)
print("%s re-projecting the converged vector: %.3f ms per projection" % (tag, tp["ms"]))
