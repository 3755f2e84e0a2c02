"""labs build: the Krylov modes under forced run-ahead stalls (SCS_HIP_PIPELINE=3) on the problem of
tests/test_minres_gpu.py::test_auto_mode_never_switches_inside_a_queued_linear_solve — who breaks?
    SCS_HIP_LIB=.../libscs_hip_labs.so python tools/dbg/mr_auto_stall.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import helpers, problem_gen as pg
from scs import _scs_hip as hip
from oracle import scs_oracle as o
K = {"z": 300, "l": 200, "q": [12, 7]}
data, p_star, _ = pg.gen_feasible(K, 320, 9, 13, lambda z, K: o.proj_cone(z, K, dual=True))
args = helpers.raw_args(data, K)
print("labs build:", hip.labs_build())
for kry in ("cg", "auto", "minres"):
    for pipe in ("1", "3", "0"):
        os.environ["SCS_HIP_KRYLOV"] = kry
        os.environ["SCS_HIP_PIPELINE"] = pipe
        for mi in (40, 400):
            sol = hip.SCS(*args, eps_abs=1e-8, eps_rel=1e-8, eps_infeas=1e-9, verbose=False, max_iters=mi).solve(False, None, None, None)
            i = sol["info"]
            print("krylov %-6s pipeline %s max_iters %4d: %-28s iter %4d cg %8d (%.1f / iter) finite %s  %s" % (
                kry, pipe, mi, i["status"], i["iter"], i["cg_iters"], i["cg_iters"] / max(i["iter"], 1), bool(np.all(np.isfinite(sol["x"]))), i["lin_sys_solver"][:60]), flush=True)
