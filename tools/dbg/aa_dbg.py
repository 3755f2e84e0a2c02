import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from scs import _scs_hip as hip
from oracle import scs_oracle as oracle
import helpers, problem_gen as pg

# 1. full solves: which problems / settings converge with adaptive_scale off
cases = []
data, K, p_star = helpers.load_problem("problems_std.npz", "std_feas_")
cases.append(("std", data, K))
Kk, n, k, seed = pg.workload("small_lp_soc")
d2, _, _ = pg.gen_feasible(Kk, n, k, seed, lambda z, K: oracle.proj_cone(z, K, dual=True))
cases.append(("small_lp_soc", d2, Kk))
K3 = {"z": 10, "l": 600, "q": [30, 12, 5], "s": [6, 3], "ep": 4, "p": [0.4, -0.7]}
d3, _, _ = pg.gen_feasible_qp(K3, 400, 7, 5, lambda z, K: oracle.proj_cone(z, K, dual=True))
cases.append(("qp_mixed", d3, K3))
for name, data, K in cases:
    args = helpers.raw_args(data, K)
    for interval in (10, 1):
        for eps in (1e-5,):
            stg = dict(eps_abs=eps, eps_rel=eps, eps_infeas=1e-9, verbose=False, adaptive_scale=False, acceleration_lookback=10,
                       acceleration_interval=interval, max_iters=5000)
            t = time.time(); got = hip.SCS(*args, **stg).solve(False, None, None, None); tg = time.time() - t
            t = time.time(); ref = oracle.OracleSCS(*args, indirect=True, **stg).solve(False); tr = time.time() - t
            gi, ri = got["info"], ref["info"]
            print(name, "interval", interval, "eps", eps, "| hip", gi["status"][:8], gi["iter"], gi["aa_stats"]["n_accept"], gi["rejected_accel_steps"], "%.1fs" % tg,
                  "| oracle", ri["status"][:8], ri["iter"], ri["aa_stats"]["n_accept"], ri["rejected_accel_steps"], "%.1fs" % tr, flush=True)

# 2. rank-deficient
for mode in ("tsqr", "gram"):
    os.environ["SCS_HIP_AA"] = mode
    for type1 in (True, False):
        dim, mem = 4000, 4
        rng = np.random.RandomState(2)
        xs, dvec = rng.randn(dim), rng.randn(dim)
        F = lambda x: xs + 0.5 * (dvec @ (x - xs)) / (dvec @ dvec) * dvec
        h, o = hip.AndersonAccelerator(dim, mem, type1=type1, regularization=0.0), oracle.OracleAA(dim, mem, type1=type1, regularization=0.0)
        x = xs + 3.0 * dvec
        for k in range(12):
            f = F(x)
            no, fo = o.apply(f, x); nh, fh = h.apply(f, x)
            print(mode, type1, k, "oracle %.3e" % no, "hip %.3e" % nh, "maxdiff %.2e" % np.abs(fo - fh).max(), o.stats()["last_rank"], h.stats()["last_rank"])
            ro, f2o, x2o = o.safeguard(F(fo), fo); rh, f2h, x2h = h.safeguard(F(fo), fo)
            x = f2o
        print(o.stats()); print(h.stats())
# 3. mem 32
for mode in ("tsqr", "gram"):
    os.environ["SCS_HIP_AA"] = mode
    for type1 in (True, False):
        dim, mem = 3000, 32
        rng = np.random.RandomState(3)
        d = rng.uniform(0.0, 0.9, dim); b = rng.randn(dim)
        F = lambda x: b + d * x + 0.05 * np.roll(x, 1)
        x = rng.randn(dim)
        h, o = hip.AndersonAccelerator(dim, mem, type1=type1), oracle.OracleAA(dim, mem, type1=type1)
        for k in range(3 * mem + 12):
            f = F(x)
            no, fo = o.apply(f, x); nh, fh = h.apply(f, x)
            if no != 0 or nh != 0:
                print(mode, type1, k, "oracle %.6e" % no, "hip %.6e" % nh, "maxdiff %.2e" % (np.abs(fo - fh).max() / max(1, np.abs(fo).max())), "res %.2e" % np.abs(x - f).max())
            ro, f2o, x2o = o.safeguard(F(fo), fo); rh, f2h, x2h = h.safeguard(F(fo), fo)
            if ro != rh: print("  safeguard differs", ro, rh)
            x = f2o
