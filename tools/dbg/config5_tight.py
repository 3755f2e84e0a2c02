"""config-5 members at eps 1e-10 (tests/test_group_gpu.py::test_config5_workload_matches_oracle_ldl in SCS_TEST_LONG mode): iterations and status per seed.
python tools/dbg/config5_tight.py [first] [count] [linear_solver]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import scs, problem_gen as pg
from scs import _scs_hip
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config5_small")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ls = sys.argv[3] if len(sys.argv) > 3 else "hip_indirect"
for sd in range(first, first + count):
    d = pg.gen_feasible(K, n, k, sd, proj)[0]
    r = scs.SCS(d, K, verbose=False, eps_abs=1e-10, eps_rel=1e-10, max_iters=60000, linear_solver=ls).solve()
    i = r["info"]
    print(sd, i["iter"], i["status"], "res_pri %.2e res_dual %.2e gap %.2e" % (i["res_pri"], i["res_dual"], i["gap"]), flush=True)
