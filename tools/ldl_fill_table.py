#!/usr/bin/env python3
"""What a SPARSE direct solver (LDL' of the KKT matrix, the reference's QDLDL / cuDSS backends: R:meson.build:238-256,374-391) would
have to store and walk for the BASELINE configurations — the numbers behind DESIGN.md §7's re-scoping of SURVEY §8 f4 to "dense
direct only" (VERDICT r05 item 7).  Symbolic phase of the oracle only (oracle/oscs_linsys.c o_lin_sys_symbolic: approximate-minimum-
degree ordering on the quotient graph + elimination tree): nnz(L) and the height of the elimination tree, on the configurations that
fit the CPU as they are and on replicas of the larger ones (same nonzeros per column, same m / n, 1/f of the rows and columns) to get
the law nnz(L) = c N^2 of a uniformly random pattern, extrapolated to full size.  CPU only (test infrastructure).

    python tools/ldl_fill_table.py > profiles/r06_ldl_fill_table.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import problem_gen as pg  # noqa: E402
from oracle import scs_oracle  # noqa: E402

# (name, m, n, nonzeros per column, pattern) of problem_gen.workload's configurations (the cone does not change the KKT pattern)
# last entry: ms per ADMM iteration of the shipped indirect path on one MI355X (profiles/r05_bench_output.json / r06: 1000 / iters per s)
CONFIGS = [
    ("config 1  LP m=4000 n=2000", 4000, 2000, 50, "uniform", None),
    ("config 5  member m=4050 n=1350", 4050, 1350, 40, "uniform", None),
    ("config 2  LP+SOC m=2e5 n=1e5", 200000, 100000, 20, "uniform", 0.38),
    ("config 3  mixed m=1e6 n=5e5", 999999, 500000, 20, "uniform", 19.3),
    ("config 4  PSD m=1.006e6 n=3.35e5", 1006000, 335000, 30, "uniform", 1.57),
    ("metric    LP+SOC m=2e6 n=1e6", 2000000, 1000000, 20, "uniform", 3.19),
    ("banded_lp (bench line) m=2e6 n=1e6", 2000000, 1000000, 20, "banded", 1.91),
]
MAX_N = int(os.environ.get("LDL_TABLE_MAX_N", "24000"))   # replicas up to this KKT order (ordering + symbolic: ~a minute at 24 000)


def pattern(m, n, k, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "banded":
        return pg.banded_sparse(m, n, max(1, int(round(k * n / m))), rng)
    return pg.random_sparse(m, n, k, rng)


def main():
    print("# tools/ldl_fill_table.py — symbolic LDL' of the KKT matrix [[rho I, A'], [A, -R_y]] under the oracle's AMD ordering (CPU, round 6)")
    print("# N = n + m; nnz(K) upper triangle incl. diagonal; height = elimination-tree height (columns that must be factored one after the other)")
    print("%-36s %5s %8s %10s %12s %9s %8s %7s" % ("configuration", "1/f", "N", "nnz(K)", "nnz(L)", "nnz(L)/N^2", "height", "sec"))
    for name, m, n, k, kind, ms_iter in CONFIGS:
        N = m + n
        fs = [1] if N <= MAX_N else sorted({max(1, int(np.ceil(N / s))) for s in (MAX_N / 4, MAX_N / 2, MAX_N)}, reverse=True)
        last = None
        for f in fs:
            mm, nn = m // f, n // f
            if kind == "banded" and f > 1:
                mm, nn = m // f, n // f
            A = pattern(mm, nn, k, kind, 1)
            t0 = time.time()
            lnz, h = scs_oracle.ldl_symbolic(A)
            dt = time.time() - t0
            NN = mm + nn
            last = (lnz, NN, h)
            print("%-36s %5d %8d %10d %12d %9.4f %8d %7.1f" % (name, f, NN, A.nnz + NN, lnz, lnz / float(NN) ** 2, h, dt), flush=True)
        lnz, NN, h = last
        if NN < N:
            if kind == "banded":
                est = lnz / NN * N      # a band: nnz(L) grows like N, the tree is a chain
                print("%-36s %5s %8d %10s %12.3g %9s %8.3g   <- full size, linear law: %.2f GB of L — and an elimination tree that is ONE chain of %.2g dependent\n"
                      "%-36s        columns: at the ~1 us a dependent step costs on the device (a 16 x 16 pivot, DESIGN §4 K9) >= %.1f s per factorisation and twice that chain\n"
                      "%-36s        per ADMM iteration for the two triangular solves, against %.2f ms per iteration of the indirect path (9 CG steps on this pattern; a cyclic-reduction / SPIKE solver could break the chain, for nothing to gain)"
                      % ("", "full", N, "", est, "", h / NN * N, est * 12 / 1e9, h / NN * N, "", h / NN * N * 1e-6, "", ms_iter))
            else:
                c = lnz / float(NN) ** 2
                est = c * float(N) ** 2
                gb = est * 12 / 1e9
                solve_ms = 2 * gb / 8000.0 * 1e3   # forward + backward substitution stream L once each, at the full 8 TB/s
                dense_order = (2 * est) ** 0.5     # the filled-in trailing block is dense: order ~ sqrt(2 nnz(L))
                fact_s = dense_order ** 3 / 3 / 78.6e12
                print("%-36s %5s %8d %10s %12.3g %9.4f %8s   <- full size, N^2 law: %.3g GB of L (fp64 + int32; HBM: 288 GB): %s;\n"
                      "%-36s        the two triangular solves of ONE ADMM iteration stream L twice: >= %.1f ms at 8 TB/s against %.2f ms per iteration of the indirect\n"
                      "%-36s        path (measured, all of it); numeric factorisation >= %.0f s at the fp64 MFMA peak (dense trailing block of order %.2g), per scale update"
                      % ("", "full", N, "", est, c, "~N", gb, "does not fit" if gb > 288 else "fits", "", solve_ms, ms_iter, "", fact_s, dense_order))
        else:
            print("%-36s %5s %8d %10s %12d %9s %8s   <- as it is: %.1f MB of L; the dense direct solver stores 8 n^2 = %.1f MB for it"
                  % ("", "full", N, "", lnz, "", "", lnz * 12 / 1e6, 8.0 * n * n / 1e6))


if __name__ == "__main__":
    main()
