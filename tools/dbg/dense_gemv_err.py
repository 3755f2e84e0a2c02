import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from scipy import sparse
from scs import _scs_hip as hip
from oracle import scs_oracle as oracle
def rand_csc(m, n, density, seed):
    rng = np.random.RandomState(seed)
    A = sparse.rand(m, n, density, format="csc", random_state=rng); A.data = rng.randn(A.nnz); A.sort_indices(); return A
for m, n, dens in [(90, 40, 0.2), (300, 65, 0.1), (600, 250, 0.03), (1500, 700, 0.01), (4050, 1350, 0.03)]:
    A = rand_csc(m, n, dens, 21 + n)
    rng = np.random.RandomState(8)
    nz = min(50, m // 3)
    diag_r = np.concatenate([np.full(n, 1e-3), np.full(nz, 0.01), np.full(m - nz, 10.0)])
    rhs = rng.randn(n + m)
    ref, _ = oracle.kkt_solve(A, None, diag_r, rhs, indirect=False)
    got = hip.kkt_solve_dense(A, None, diag_r, rhs)
    G = (sparse.diags(diag_r[:n]) + A.T @ sparse.diags(1.0 / diag_r[n:]) @ A).toarray()
    print("m %5d n %5d  %s  max err / max|ref| = %.3e   cond(G) = %.2e" % (m, n, os.environ.get("SCS_HIP_DENSE_GEMV", "half"), np.abs(got - ref).max() / np.abs(ref).max(), np.linalg.cond(G)))
