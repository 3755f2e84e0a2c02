#!/usr/bin/env python3
"""K3 (QP path, Gp = P p) — what the full symmetric CSR `Pf` costs against the alternative VERDICT r05 item 4 names: the stored upper
triangle walked twice without float atomics (a row-ordered pass y = U x and a column-ordered pass y += strict(U)' x, fixed order).
Both orders need their own copy of the values (a pass that gathers the values of the other order through a permutation reads a 128-byte
line per 8-byte value), so the two-pass variant streams the bytes of Pf in two launches; this tool measures both with the shipped
kernels (scs_hip_spmv_bench: HIP events around `reps` launches, inputs resident) on the bench's `target_qp` P.

    python tools/k3_ab.py            # on the GPU box; prints one table (profiles/r06_k3.txt)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "scs-python_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from scipy import sparse  # noqa: E402
import problem_gen as pg  # noqa: E402
from scs import _scs_hip  # noqa: E402

HBM = 8000.0


def main():
    K, n, k, seed = pg.workload("target_qp")
    t0 = time.time()
    data, _, _ = pg.gen_feasible_qp(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True), b_per_col=pg.workload_qp("target_qp"))
    U = data["P"].tocsc()
    U.sort_indices()
    Ls = sparse.triu(U, 1).T.tocsc()   # strict lower triangle = the transposed strict upper one
    Ls.sort_indices()
    Pf = (U + Ls).tocsc()
    Pf.sort_indices()
    print("target_qp: n = %d, nnz(triu P) = %d, nnz(Pf) = %d  (generated in %.1f s)" % (n, U.nnz, Pf.nnz, time.time() - t0))
    alg = 12 * U.nnz + 4 * (n + 1) + 16 * n
    rows = []
    for name, M in (("Pf: full symmetric CSR, one launch (shipped)", Pf), ("pass 1: y = U x (rows of the stored triangle)", U),
                    ("pass 2: y += strict(U)' x (columns of the stored triangle)", Ls)):
        ms = min(_scs_hip.spmv_bench(M, transpose=False, reps=30) for _ in range(3))
        streamed = 12 * M.nnz + 4 * (n + 1) + 16 * n
        rows.append((name, M.nnz, ms, streamed))
    x = np.random.RandomState(1).randn(n)
    one = _scs_hip.spmv(Pf, x)
    two = _scs_hip.spmv(U, x) + _scs_hip.spmv(Ls, x)
    print("%-62s %10s %9s %10s %8s" % ("variant", "nnz", "us", "GB/s strm", "frac alg"))
    for name, nnz, ms, streamed in rows:
        print("%-62s %10d %9.1f %10.0f %8.3f" % (name, nnz, ms * 1e3, streamed / (ms * 1e-3) / 1e9, alg / (ms * 1e-3) / 1e9 / HBM))
    two_ms = rows[1][2] + rows[2][2]
    print("%-62s %10d %9.1f %10s %8.3f" % ("two passes together (+ one more launch)", rows[1][1] + rows[2][1], two_ms * 1e3, "", alg / (two_ms * 1e-3) / 1e9 / HBM))
    print("algorithmic bytes (SURVEY 2.1 K3: 12 B per STORED entry + row pointers + p + Gp) = %.1f MB; Pf streams %.1f MB" % (alg / 1e6, rows[0][3] / 1e6))
    print("max |one launch - two passes| / max|y| = %.2e (different association of the two triangles' sums)" % (np.abs(one - two).max() / np.abs(one).max()))


if __name__ == "__main__":
    main()
