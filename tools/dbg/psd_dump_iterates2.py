"""Round 6 lab input for the sign-split sweeps of K9: the matrices config 4 REALLY projects at consecutive ADMM
iterations in every phase of a solve (cold, the window before the refinement gate opens, across an Anderson step, steady state).
z = y - s of solves stopped at k, k+1, ... (Moreau: Pi_+(z) = y; solves are bit-deterministic, so they share their first k iterations).
    python tools/dbg/psd_dump_iterates2.py gpurun_out/psd_iterates2.npz      (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import scs
from scs import _scs_hip
import problem_gen as pg

proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config4_psd")
data, _, _ = pg.gen_feasible(K, n, k, seed, proj)
o, d = K["l"], 200 * 201 // 2
blocks = [0, 9, 18, 27, 36, 45]
out = {"blocks": np.array(blocks)}
starts = (21, 61, 101, 108, 131, 161, 201, 241, 301)
out["starts"] = np.array(starts)
for k0 in starts:
    for it in range(k0, k0 + 4):
        sol = scs.SCS(data, K, verbose=False, max_iters=it, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, acceleration_lookback=10,
                      linear_solver="hip_indirect").solve(warm_start=False)
        z = sol["y"] - sol["s"]
        out["z_%d" % it] = np.stack([z[o + b * d:o + (b + 1) * d] for b in blocks])
        print(it, sol["info"]["iter"], sol["info"]["res_pri"], sol["info"]["res_dual"], sol["info"]["aa_stats"]["n_accept"], flush=True)
np.savez_compressed(sys.argv[1], **out)
