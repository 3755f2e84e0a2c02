"""stand-alone products of the bench matrix through the C-ABI (tools/prof_spmv.sh profiles this)"""
import os, sys, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scs-python_amd")); sys.path.insert(0, ROOT)
from scs import _scs_hip as hip
import problem_gen as pg
rng = np.random.default_rng(0)
A = pg.random_sparse(2000000, 1000000, 20, rng)
nnz = A.nnz
for trans in (False, True):
    ms = hip.spmv_bench(A, transpose=trans, reps=30)
    rows, cols = (1000000, 2000000) if trans else (2000000, 1000000)
    b = 12*nnz + 4*(rows+1) + 8*cols + 8*rows
    print("trans" if trans else "plain", "ms %.4f" % ms, "GB/s %.0f" % (b/ms/1e6))
