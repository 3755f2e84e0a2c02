"""how far do iteration / Anderson counters of the HIP path drift from the oracle's CPU-CG variant on the mixed-cone QP
of tests/test_aa_gpu.py, over seeds — and how far does the oracle drift from ITSELF (CG vs LDL' linear solves)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from scs import _scs_hip as hip
from oracle import scs_oracle as oracle
import problem_gen as pg
import helpers

proj = lambda z, K: oracle.proj_cone(z, K, dual=True)
K2 = {"z": 10, "l": 600, "q": [30, 12, 5], "s": [6, 3], "ep": 4, "p": [0.4, -0.7]}
STG = dict(eps_infeas=1e-9, verbose=False, adaptive_scale=False, acceleration_lookback=10, acceleration_interval=10,
           eps_abs=1e-5, eps_rel=1e-5, max_iters=20000)
for type1 in (True, False):
    for seed in range(5, 13):
        d, p, _ = pg.gen_feasible_qp(K2, 400, 7, seed, proj)
        args = helpers.raw_args(d, K2)
        stg = dict(STG, acceleration_type_1=type1)
        g = hip.SCS(*args, **stg).solve(False, None, None, None)["info"]
        r = oracle.OracleSCS(*args, indirect=True, **stg).solve(False)["info"]
        q = oracle.OracleSCS(*args, indirect=False, **stg).solve(False)["info"]
        print("type1=%d seed %2d: iters hip %5d oracle-cg %5d oracle-ldl %5d | hip vs cg %+.2f, ldl vs cg %+.2f | accepts %d / %d / %d | rejects %d / %d / %d"
              % (type1, seed, g["iter"], r["iter"], q["iter"], (g["iter"] - r["iter"]) / r["iter"], (q["iter"] - r["iter"]) / r["iter"],
                 g["aa_stats"]["n_accept"], r["aa_stats"]["n_accept"], q["aa_stats"]["n_accept"],
                 g["rejected_accel_steps"], r["rejected_accel_steps"], q["rejected_accel_steps"]))
