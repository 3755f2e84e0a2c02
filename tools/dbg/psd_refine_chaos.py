"""Round 5: iteration counts of whole config-4 solves (50 PSD cones of order 200, Anderson acceleration on) under perturbations that
have nothing to do with the projection (alpha 1.5 + 1e-9 / + 1e-6, scale 0.1000001, alpha 1.49), with K9's refinement stage
on and off: is the spread BETWEEN the two modes inside the spread each mode shows by itself?  (cf. tools/dbg/psd_tol_chaos.py)
    python tools/dbg/psd_refine_chaos.py [eps]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
if len(sys.argv) > 2:  # child: one mode
    import scs, problem_gen as pg
    from scs import _scs_hip
    eps = float(sys.argv[1])
    K, n, k, seed = pg.workload("config4_psd")
    d = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))[0]
    row = []
    for kw in (dict(), dict(alpha=1.5 + 1e-9), dict(alpha=1.5 + 1e-6), dict(scale=0.1000001), dict(alpha=1.49)):
        s = scs.SCS(d, K, verbose=False, eps_abs=eps, eps_rel=eps, max_iters=6000, **kw)
        r = s.solve()
        st = s._solver._psd_refine_stats()
        row.append("%d (%s, pobj %.6f, refined %.0f, sent back %.2f)" % (r["info"]["iter"], r["info"]["status"], r["info"]["pobj"], st[:, 0].mean(), st[:, 1].mean()))
    print("REFINE=%s eps %g: iterations as is / alpha+1e-9 / alpha+1e-6 / scale+1e-7 / alpha 1.49:\n   " % (os.environ.get("SCS_HIP_PSD_REFINE", "1"), eps) + "\n   ".join(row), flush=True)
else:
    eps = sys.argv[1] if len(sys.argv) > 1 else "1e-4"
    for mode in ("1", "0"):
        env = dict(os.environ, SCS_HIP_PSD_REFINE=mode)
        subprocess.run([sys.executable, __file__, eps, "child"], env=env, check=True)
