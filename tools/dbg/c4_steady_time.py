"""config 4: time of iterations [0,105) and [105,225) of one solve, with the cooperative / ordinary launch of the multi-CU sweep kernel"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import scs, problem_gen as pg
from scs import _scs_hip
K, n, k, seed = pg.workload("config4_psd")
d = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))[0]
for rep in range(2):
    s = scs.SCS(d, K, verbose=False, eps_abs=0., eps_rel=0., eps_infeas=0., max_iters=225, acceleration_lookback=int(os.environ.get("LOOKBACK", "10")))
    s._solver._set_mark(105)
    r = s.solve()
    mk = s._solver._get_mark()
    i = r["info"]
    print("COOP=%s AA=%s rep %d: [0,105) %.1f ms = %.0f it/s   [105,225) %.1f ms = %.0f it/s   lin %.0f cone %.0f accel %.0f ms  aa %s" % (
        os.environ.get("SCS_HIP_PSD_COOP", "1"), os.environ.get("LOOKBACK", "10"), rep, mk["ms"], 105e3 / mk["ms"], i["solve_time"] - mk["ms"], 120e3 / (i["solve_time"] - mk["ms"]),
        i["lin_sys_time"], i["cone_time"], i["accel_time"], i["aa_stats"]))
