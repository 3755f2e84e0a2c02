"""Round 5 lab input for the GEMM-only refinement of K9's warm-started eigenbasis: the matrices config 4 REALLY projects at
CONSECUTIVE ADMM iterations (z = y - s of solves stopped at k, k+1, ...; Moreau: Pi_+(z) = y, Pi_-(z) = -s, so z is the projection's
input up to the iterate's scaling).  Solves are bit-deterministic, so solves stopped at k and k+1 share their first k iterations.
    python tools/dbg/psd_dump_iterates.py gpurun_out/psd_iterates.npz      (GPU box)"""
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
blocks = list(range(0, 50, 8))
out = {"blocks": np.array(blocks)}
for k0 in (100, 300, 600):
    for it in range(k0, k0 + 12):
        sol = scs.SCS(data, K, verbose=False, max_iters=it, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0).solve(warm_start=False)
        z = sol["y"] - sol["s"]
        out["z_%d" % it] = np.stack([z[o + b * d:o + (b + 1) * d] for b in blocks])
        print(it, sol["info"]["iter"], sol["info"]["res_pri"], sol["info"]["res_dual"], flush=True)
np.savez(sys.argv[1], **out)
