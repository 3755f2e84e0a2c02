"""SpMV time of the layout scs_init would pick against the plain CSR-stream kernel (SCS_HIP_SLAB=0, child process) on adversarial
sparsity patterns: any pattern where the chosen layout is several times SLOWER than the fallback is a cliff (round 3 found one: a
budget row left whole in the passes).  usage: python tools/dbg/pattern_cliffs.py [child <name>]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from scipy import sparse as sp


def make(name):
    import problem_gen as pg
    rng = np.random.default_rng(5)
    if name == "uniform_1e6x5e5":
        return pg.random_sparse(1000000, 500000, 10, rng)
    if name == "dense_row_and_col_small":
        A = pg.random_sparse(140000, 70000, 16, rng).tolil()
        A[5, :] = rng.standard_normal(70000); A[:, 9] = rng.standard_normal(140000).reshape(-1, 1)
        return sp.csc_matrix(A)
    if name == "dense_row_and_col_big":
        A = pg.random_sparse(1000000, 500000, 10, rng).tocoo()
        r = np.concatenate([A.row, np.full(500000, 7), np.arange(1000000)]); c = np.concatenate([A.col, np.arange(500000), np.full(1000000, 11)])
        v = np.concatenate([A.data, rng.standard_normal(1500000)])
        return sp.csc_matrix(sp.coo_matrix((v, (r, c)), shape=(1000000, 500000)))
    if name == "half_the_nonzeros_in_20_rows":
        A = pg.random_sparse(1000000, 500000, 5, rng).tocoo()
        rr = np.repeat(np.arange(20) * 50000 + 3, 125000); cc = np.tile(rng.choice(500000, 125000, replace=False), 20)
        return sp.csc_matrix(sp.coo_matrix((np.concatenate([A.data, rng.standard_normal(rr.size)]), (np.concatenate([A.row, rr]), np.concatenate([A.col, cc]))),
                                           shape=(1000000, 500000)))
    if name == "many_rows_of_1500":   # rows a 12/13-bit count field holds: whole rows with long runs
        return sp.csc_matrix(sp.random(40000, 200000, density=1500 / 200000, random_state=3, format="csr", data_rvs=rng.standard_normal))
    if name == "powerlaw_2e6":
        return pg.powerlaw_sparse(2000000, 1000000, 10, rng)
    if name == "clustered_long_rows":   # long rows whose columns are CONSECUTIVE (a piece's nonzeros all in one pass)
        A = pg.random_sparse(1000000, 500000, 8, rng).tocoo()
        rr = np.repeat(np.arange(200) * 5000 + 1, 5000); cc = (np.tile(np.arange(5000), 200) + np.repeat(rng.integers(0, 490000, 200), 5000))
        return sp.csc_matrix(sp.coo_matrix((np.concatenate([A.data, rng.standard_normal(rr.size)]), (np.concatenate([A.row, rr]), np.concatenate([A.col, cc]))),
                                           shape=(1000000, 500000)))
    raise SystemExit("unknown pattern " + name)


NAMES = ["uniform_1e6x5e5", "dense_row_and_col_small", "dense_row_and_col_big", "half_the_nonzeros_in_20_rows", "many_rows_of_1500", "powerlaw_2e6",
         "clustered_long_rows"]
if len(sys.argv) > 2 and sys.argv[1] == "child":
    from scs import _scs_hip as hip
    A = make(sys.argv[2]); A.sum_duplicates(); A.sort_indices()
    out = {"shape": A.shape, "nnz": int(A.nnz), "max_row": int(np.diff(A.tocsr().indptr).max()), "max_col": int(np.diff(A.indptr).max())}
    for tr in (False, True):
        out["A'" if tr else "A"] = round(hip.spmv_bench(A, transpose=tr, reps=10) * 1e3, 1)
    print(json.dumps(out))
else:
    for name in NAMES:
        row = {}
        for tag, env in (("chosen", {}), ("csr_stream", {"SCS_HIP_SLAB": "0"})):
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", name], env=e, capture_output=True, text=True, timeout=600)
            js = [l for l in r.stdout.splitlines() if l.startswith("{")]
            row[tag] = json.loads(js[-1]) if js else {"error": r.stderr[-300:]}
        c, f = row["chosen"], row["csr_stream"]
        if "error" in c or "error" in f:
            print(name, row); continue
        print("%-30s nnz %9d max row %7d max col %7d | chosen layout: A x %8.1f us, A' y %8.1f us | CSR-stream: %8.1f / %8.1f us | ratio %.2f / %.2f" % (
            name, c["nnz"], c["max_row"], c["max_col"], c["A"], c["A'"], f["A"], f["A'"], c["A"] / f["A"], c["A'"] / f["A'"]), flush=True)
