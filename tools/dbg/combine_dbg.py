import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from scs import _scs_hip as hip
import helpers, problem_gen as pg
K = {"l": 400000}
data, p_star, _ = pg.gen_feasible(K, 300000, 6, 8, lambda z, K: hip.proj_cone(z, K, dual=True))
print("p*", p_star)
for name, env in (("default", {}), ("split0", {"SCS_HIP_CS_SPLIT": "0"}), ("sched1", {"SCS_HIP_CS_SCHED": "1"}), ("stream", {"SCS_HIP_SLAB": "0"}),
                  ("combine11", {"SCS_HIP_CS_COMBINE": "1", "SCS_HIP_CS_SPLIT_A": "1", "SCS_HIP_CS_SPLIT_AT": "2"}),
                  ("combine24", {"SCS_HIP_CS_COMBINE": "1", "SCS_HIP_CS_SPLIT_A": "2", "SCS_HIP_CS_SPLIT_AT": "4"})):
    for k in ("SCS_HIP_CS_COMBINE", "SCS_HIP_CS_SPLIT_A", "SCS_HIP_CS_SPLIT_AT", "SCS_HIP_CS_SPLIT", "SCS_HIP_CS_SCHED", "SCS_HIP_SLAB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for eps in (1e-5, 1e-7):
        t = time.time()
        s = hip.SCS(*helpers.raw_args(data, K), eps_abs=eps, eps_rel=eps, verbose=False, max_iters=20000).solve(False, None, None, None)
        i = s["info"]
        print(name, eps, i["status"][:12], "iters", i["iter"], "cg", i["cg_iters"], "pobj %.6f" % i["pobj"], "scale_updates", i["scale_updates"], "aa acc", i["aa_stats"]["n_accept"], "rej", i["rejected_accel_steps"], "%.1fs" % (time.time() - t), i["lin_sys_solver"][:60], flush=True)
