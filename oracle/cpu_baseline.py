"""bench.py's CPU legs, each run as a CHILD process so that they can be capped by wall clock and can pick the
OpenMP build — TEST INFRASTRUCTURE (oracle/), never the product path.

    python oracle/cpu_baseline.py cg  <workload> <iters> <threads>      CPU-CG variant of the oracle, `threads` OpenMP threads
    python oracle/cpu_baseline.py ldl <m> <n> <nnz_per_col> <seed> <iters>   sparse LDL' direct variant on a random LP of that size

Prints ONE JSON line.  The instance is generated here with problem_gen (the oracle's own CPU cone projection), so the
child needs neither the GPU nor the parent's data.  `threads` > 1 loads liboscs_omp.so (oracle/Makefile): row- /
column-parallel mat-vecs, tree reductions — a timing build; the checker the tests use stays sequential.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    if mode == "cg":
        workload, iters = sys.argv[2], int(sys.argv[3])
        sweep = [int(t) for t in sys.argv[4].split(",")]   # one thread count, or a sweep "64,128,256" (one line per count, the best last)
        threads = sweep[0]
        if sweep != [1]:
            os.environ["OSCS_LIB"] = "omp"
            os.environ["OMP_NUM_THREADS"] = str(max(sweep))
            os.environ.setdefault("OMP_PROC_BIND", "spread")
            os.environ.setdefault("OMP_PLACES", "threads")
        from oracle import scs_oracle
        import problem_gen as pg
        K, n, k, seed = pg.workload(workload)
        t = time.perf_counter()
        data, _, _ = pg.gen_feasible(K, n, k, seed, lambda z, KK: scs_oracle.proj_cone(z, KK, dual=True),
                                     pattern=pg.workload_pattern(workload))
        tgen = time.perf_counter() - t
        lines = []
        for threads in sweep:
            if sweep != [1]:
                scs_oracle.lib().oscs_set_num_threads(threads)  # the workspace's pages are first touched by these threads
            r = scs_oracle.solve(data, K, indirect=True, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False,
                                 acceleration_lookback=10, max_iters=iters)
            ms = r["info"]["solve_time"]
            lines.append({"mode": "cg", "workload": workload, "threads": threads, "iters": iters, "solve_s": ms * 1e-3,
                          "iters_per_s": iters / (ms * 1e-3), "cg_steps": r["info"]["cg_iters"], "gen_s": tgen,
                          "setup_s": r["info"]["setup_time"] * 1e-3})
            print(json.dumps(lines[-1]))
            sys.stdout.flush()
        if len(lines) > 1:
            best = max(lines, key=lambda d: d["iters_per_s"])
            print(json.dumps(dict(best, sweep=[[d["threads"], round(d["iters_per_s"], 3)] for d in lines])))
    elif mode == "ldl":
        m, n, k, seed, iters = (int(a) for a in sys.argv[2:7])
        from oracle import scs_oracle
        import problem_gen as pg
        K = {"l": m}
        data, _, _ = pg.gen_feasible(K, n, k, seed, lambda z, KK: scs_oracle.proj_cone(z, KK, dual=True))
        r = scs_oracle.solve(data, K, indirect=False, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False,
                             acceleration_lookback=10, max_iters=iters)
        ms = r["info"]["solve_time"]
        lss = r["info"].get("lin_sys_solver", "")
        nnz_l = int(lss.split("nnz(L)=")[1]) if "nnz(L)=" in lss else None
        print(json.dumps({"mode": "ldl", "m": m, "n": n, "nnz": int(data["A"].nnz), "nnz_L": nnz_l, "iters": iters,
                          "factorization_s": r["info"]["setup_time"] * 1e-3, "solve_s": ms * 1e-3,
                          "iters_per_s": iters / (ms * 1e-3)}))
    else:
        raise SystemExit("unknown mode %r" % mode)
    sys.stdout.flush()


if __name__ == "__main__":
    main()
