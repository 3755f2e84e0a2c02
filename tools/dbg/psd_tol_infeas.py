"""infeasible / unbounded goldens (with PSD cones) under the residual-tied PSD stopping level: iterations to the certificate"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import scs
import helpers
mode = os.environ.get("SCS_HIP_PSD_TOL", "adaptive") + " k=" + os.environ.get("SCS_HIP_PSD_TOL_K", "dflt")
for fname, prefix in (("problems_std.npz", "std_infeas_"), ("problems_std.npz", "std_unbdd_"), ("problems_sdp.npz", "infeas0_"), ("problems_sdp.npz", "infeas1_"),
                      ("problems_sdp.npz", "unbdd0_"), ("problems_sdp.npz", "unbdd1_")):
    data, K, _ = helpers.load_problem(fname, prefix)
    for kw in (dict(), dict(acceleration_type_1=False, acceleration_interval=1, acceleration_lookback=5), dict(acceleration_lookback=0)):
        t = time.time()
        sol = scs.SCS(data, K, verbose=False, max_iters=20000, **kw).solve()
        i = sol["info"]
        print("%-16s %s%-12s %-40s %-45s iters %6d res_infeas %.1e unbdd %.1e/%.1e %.1fs" % (mode, fname[9:12], prefix, str(kw)[:40], i["status"], i["iter"],
              i["res_infeas"], i["res_unbdd_a"], i["res_unbdd_p"], time.time() - t))
