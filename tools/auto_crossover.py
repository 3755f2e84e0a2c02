#!/usr/bin/env python3
"""Where `LinearSolver.AUTO` should stop preferring the dense direct solver (scs/__init__.py `_resolve_auto`): whole solves
(scs.SCS(...) + solve(), default settings) of random LP+SOC programs of growing order with both linear solvers of the device.
Prints one table (DESIGN §4 "AUTO", profiles/r06_auto_crossover.txt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "scs-python_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import scs  # noqa: E402
import problem_gen as pg  # noqa: E402
from scs import _scs_hip  # noqa: E402


def main():
    proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)  # noqa: E731
    print("%6s %7s %9s | %-13s %8s %8s %7s %9s | %-13s %8s %8s %7s %9s" % (
        "n", "m", "nnz", "solver", "init s", "solve s", "iters", "ms/iter", "solver", "init s", "solve s", "iters", "ms/iter"))
    for n in (256, 1350, 2048, 4096, 6144, 8192):
        m = 3 * n
        K = {"l": 2 * n, "q": [16] * (n // 16)}
        data, p_star, _ = pg.gen_feasible(K, n, 30, 77 + n, proj)
        row = "%6d %7d %9d" % (n, m, data["A"].nnz)
        for ls in ("hip_dense", "hip_indirect"):
            best = None
            for rep in range(2):
                t0 = time.perf_counter()
                sv = scs.SCS(data, K, verbose=False, linear_solver=ls)
                t1 = time.perf_counter()
                sol = sv.solve()
                t2 = time.perf_counter()
                del sv
                rec = (t1 - t0, t2 - t1, sol["info"]["iter"], sol["info"]["status"])
                if best is None or rec[0] + rec[1] < best[0] + best[1]:
                    best = rec
            assert best[3] == "solved", (n, ls, best)
            assert abs(sol["info"]["pobj"] - p_star) < 5e-3 * max(1.0, abs(p_star)), (n, ls, sol["info"]["pobj"], p_star)
            row += " | %-13s %8.3f %8.3f %7d %9.4f" % (ls, best[0], best[1], best[2], 1e3 * best[1] / max(best[2], 1))
        print(row, flush=True)


if __name__ == "__main__":
    main()
