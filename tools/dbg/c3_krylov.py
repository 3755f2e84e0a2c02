"""config 3 (BASELINE configs[2]: z / l / box / q / ep / ed / p, m = 999 999), a whole solve to eps 1e-4 with PCG and with MINRES on the
un-eliminated zero-cone block: ADMM iterations, Krylov steps, wall time — the two must be compared on the WHOLE solve, not per step:
at equal residual tolerance the two methods leave different error, and the ADMM loop feels the error.
    python tools/dbg/c3_krylov.py [scale_down]"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
if len(sys.argv) > 2:
    import numpy as np
    import scs, problem_gen as pg
    from scs import _scs_hip
    K, n, k, seed = pg.workload("config3_mixed")
    data, p_star, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))
    s = scs.SCS(data, K, verbose=False, eps_abs=1e-4, eps_rel=1e-4, max_iters=int(os.environ.get("MAXIT", "20000")))
    t0 = time.time()
    r = s.solve()
    t1 = time.time()
    i = r["info"]
    print("KRYLOV=%s TOLF=%s: %s, %d ADMM iterations, %d Krylov steps (%.1f per iteration), solve %.2f s = %.1f it/s; pobj %.6f (p* %.6f); %s" % (
        os.environ.get("SCS_HIP_KRYLOV", "auto"), os.environ.get("SCS_HIP_MR_TOLF", "1"), i["status"], i["iter"], i["cg_iters"], i["cg_iters"] / max(i["iter"], 1),
        i["solve_time"] / 1e3, i["iter"] / (i["solve_time"] / 1e3), i["pobj"], p_star, i["lin_sys_solver"]), flush=True)
else:
    for mode, tolf in (("cg", "1"), ("minres", "1"), ("minres", "0.1"), ("minres", "0.01")):
        env = dict(os.environ, SCS_HIP_KRYLOV=mode, SCS_HIP_MR_TOLF=tolf)
        subprocess.run([sys.executable, __file__, "x", "child"], env=env, check=False)
