"""tests/test_hip_parity.py::test_run_ahead_loop_bit_identical[sdp] with SCS_HIP_DEBUG=tol: per-iteration stopping level in both loop modes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scs-python_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import numpy as np
import problem_gen as pg, helpers, scs_oracle as oracle
from scs import _scs_hip as hip
proj = lambda z, K: oracle.proj_cone(z, K, dual=True)
K = {"l": 30, "s": [40, 12], "cs": [5]}
data, _, _ = pg.gen_feasible_qp(K, pg.cone_dims(K) + 2, 6, 21, proj)
args = helpers.raw_args(data, K)
mode = sys.argv[1]
os.environ["SCS_HIP_PIPELINE"] = mode
sol = hip.SCS(*args, eps_abs=1e-7, eps_rel=1e-7, eps_infeas=1e-9, max_iters=600, verbose=False).solve(False, None, None, None)
print("mode", mode, sol["info"]["iter"], sol["info"]["status"], sol["info"]["cg_iters"] if "cg_iters" in sol["info"] else "", file=sys.stderr)
