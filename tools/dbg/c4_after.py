"""why is config 4's steady window 2x slower inside a default bench run?  time it fresh, then after other workloads ran in this process"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import scs, problem_gen as pg, torch
from scs import _scs_hip
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config4_psd")
d = pg.gen_feasible(K, n, k, seed, proj)[0]
def c4(tag):
    s = scs.SCS(d, K, verbose=False, eps_abs=0., eps_rel=0., eps_infeas=0., max_iters=225, acceleration_lookback=10)
    s._solver._set_mark(105)
    torch.cuda.synchronize(); time.sleep(0.4)
    r = s.solve(); mk = s._solver._get_mark(); i = r["info"]
    print("%-44s [0,105) %.0f it/s  [105,225) %.0f it/s" % (tag, 105e3 / mk["ms"], 120e3 / (i["solve_time"] - mk["ms"])), flush=True)
c4("fresh process")
what = sys.argv[1]
if what in ("batch", "all"):
    Kb, nb_, kb_, seedb = pg.workload("config5_small")
    ss = [scs.SCS(pg.gen_feasible(Kb, nb_, kb_, seedb + i, proj)[0], Kb, verbose=False, linear_solver="hip_dense") for i in range(128)]
    scs.solve_batch(ss); del ss
    c4("after a dense batch of 128")
if what in ("big", "all"):
    Kt, nt, kt, st = pg.workload("target_lp_soc")
    dt = pg.gen_feasible(Kt, nt, kt, st, proj)[0]
    scs.SCS(dt, Kt, verbose=False, max_iters=20, eps_abs=0., eps_rel=0.).solve(); del dt
    c4("after the metric workload (20 iterations)")
if what in ("c3", "all"):
    K3, n3, k3, s3 = pg.workload("config3_mixed")
    d3 = pg.gen_feasible(K3, n3, k3, s3, proj)[0]
    scs.SCS(d3, K3, verbose=False, max_iters=5, eps_abs=0., eps_rel=0.).solve(); del d3
    c4("after config 3 (5 iterations)")
if what in ("c2", "all"):
    K2, n2, k2, s2 = pg.workload("config2_lp_soc")
    d2 = pg.gen_feasible(K2, n2, k2, s2, proj)[0]
    scs.SCS(d2, K2, verbose=False, max_iters=100, eps_abs=0., eps_rel=0.).solve(); del d2
    c4("after config 2 (100 iterations)")
if what in ("streams",):
    Kb, nb_, kb_, seedb = pg.workload("small_lp_soc")
    db = pg.gen_feasible(Kb, nb_, kb_, seedb, proj)[0]
    ss = [scs.SCS(db, Kb, verbose=False) for i in range(40)]
    for s_ in ss: s_.solve()
    del ss
    c4("after 40 simultaneously live workspaces")
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=16) as pool:
        ss = list(pool.map(lambda i: scs.SCS(db, Kb, verbose=False), range(64)))
    scs.solve_batch(ss); del ss
    c4("after 64 workspaces set up by 16 threads")
