"""iteration counts of the SDP goldens at eps 1e-9: how much do they move under a perturbation that has nothing to do with the
PSD stopping level (alpha 1.5 -> 1.5 + 1e-9 / + 1e-6, scale 0.1 -> 0.1000001)?  Control for tools/dbg/psd_tol_effect.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import scs
import helpers
mode = os.environ.get("SCS_HIP_PSD_TOL", "adaptive") + " k=" + os.environ.get("SCS_HIP_PSD_TOL_K", "dflt")
for prefix in ("feas0_", "feas1_", "feas2_"):
    data, K, p_star = helpers.load_problem("problems_sdp.npz", prefix)
    row = []
    for kw in (dict(), dict(alpha=1.5 + 1e-9), dict(alpha=1.5 + 1e-6), dict(scale=0.1000001), dict(alpha=1.49)):
        sol = scs.SCS(data, K, verbose=False, eps_abs=1e-9, eps_rel=1e-9, **kw).solve()
        row.append(sol["info"]["iter"] if sol["info"]["status"] == "solved" else -1)
    print("%-22s sdp %s: iterations (as is, alpha+1e-9, alpha+1e-6, scale+1e-7, alpha 1.49) = %s" % (mode, prefix, row))
