"""Synthetic random cone programs with a KNOWN optimal value (bench + tests).

Own implementation of the construction used by the reference's generator
`gen_feasible` (R:test/gen_random_cone_prob.py:9-24):

    z ~ N(0,1)^m;  y = Pi_{K*}(z);  s = y - z (= Pi_K(-z));
    A sparse with N(0,1) values;  x ~ N(0,1)^n;  c = -A'y;  b = A x + s;  p* = c'x

so (x, y, s) is a primal-dual optimal triple by construction.  The cone
projection is injected (`proj_dual`) so that tests can use the oracle and
bench.py can use the HIP kernels; nothing here imports oracle/ or the product.
The sparsity pattern is drawn with a fixed number of nonzeros per column
(uniform random rows), which scales to nnz ~ 2e7 in seconds; the reference uses
scipy.sparse.rand(density) — same distribution family, different RNG stream
("the build's generator need not replicate the stream").
"""
import numpy as np
from scipy import sparse


def cone_dims(K):
    m = int(K.get("z", 0)) + int(K.get("l", 0))
    nb = len(K.get("bu", []))
    m += nb + 1 if nb else 0
    m += int(sum(K.get("q", [])))
    m += int(sum(int(s) * (int(s) + 1) // 2 for s in K.get("s", [])))
    m += int(sum(int(s) * int(s) for s in K.get("cs", [])))
    m += 3 * (int(K.get("ep", 0)) + int(K.get("ed", 0)) + len(K.get("p", [])))
    return m


def random_sparse(m, n, nnz_per_col, rng):
    """m x n CSC, `nnz_per_col` uniformly random rows per column (duplicates merged), N(0,1) values."""
    k = int(nnz_per_col)
    rows = rng.integers(0, m, size=(n, k), dtype=np.int64)
    rows.sort(axis=1)
    vals = rng.standard_normal(size=(n, k))
    # merge duplicates inside a column by zeroing the repeat and summing into the first
    dup = np.zeros((n, k), dtype=bool)
    dup[:, 1:] = rows[:, 1:] == rows[:, :-1]
    if dup.any():
        A = sparse.csc_matrix((vals.ravel(), rows.ravel().astype(np.int32),
                               np.arange(0, n * k + 1, k, dtype=np.int64)), shape=(m, n))
        A.sum_duplicates()
        A.sort_indices()
        return A
    return sparse.csc_matrix((vals.ravel(), rows.ravel().astype(np.int32),
                              np.arange(0, n * k + 1, k, dtype=np.int32)), shape=(m, n))


def powerlaw_sparse(m, n, nnz_per_row, rng, shape=1.3, cap=20000):
    """m x n CSC whose ROW lengths are heavy-tailed (Pareto, mean ~ nnz_per_row, capped), columns uniform: a few rows
    with thousands of nonzeros next to many short ones (budget / coupling constraints of real LPs look like this)."""
    raw = rng.pareto(shape, m) + 0.05
    lens = np.minimum(np.maximum((raw * (nnz_per_row / raw.mean())).astype(np.int64), 1), min(cap, n))
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    cols = rng.integers(0, n, size=rows.size, dtype=np.int64)
    A = sparse.csc_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    A.sort_indices()
    return A


def banded_sparse(m, n, nnz_per_row, rng):
    """m x n CSC with a band of nnz_per_row entries around the (stretched) diagonal: row i touches columns
    floor(i n / m) + (-k/2 .. k/2) — perfect gather locality, the opposite extreme of the uniform pattern."""
    k = int(nnz_per_row)
    base = (np.arange(m, dtype=np.int64) * n) // m
    cols = base[:, None] + np.arange(-(k // 2), k - k // 2, dtype=np.int64)[None, :]
    ok = (cols >= 0) & (cols < n)
    rows = np.broadcast_to(np.arange(m, dtype=np.int64)[:, None], cols.shape)[ok]
    A = sparse.csc_matrix((rng.standard_normal(int(ok.sum())), (rows, cols[ok])), shape=(m, n))
    A.sort_indices()
    return A


def gen_feasible(K, n, nnz_per_col, seed, proj_dual, pattern="uniform"):
    """Returns (data, p_star, (x, y, s)).  proj_dual(z, K) must return Pi_{K*}(z).
    pattern: "uniform" (nnz_per_col random rows per column), "powerlaw" / "banded" (nnz_per_col * n / m per row)."""
    rng = np.random.default_rng(seed)
    m = cone_dims(K)
    z = rng.standard_normal(m)
    y = np.asarray(proj_dual(z, K), dtype=np.float64)
    s = y - z
    if pattern == "uniform":
        A = random_sparse(m, n, nnz_per_col, rng)
    elif pattern == "powerlaw":
        A = powerlaw_sparse(m, n, nnz_per_col * n / m, rng)
    elif pattern == "banded":
        A = banded_sparse(m, n, max(1, int(round(nnz_per_col * n / m))), rng)
    else:
        raise KeyError(pattern)
    x = rng.standard_normal(n)
    c = -(A.T @ y)
    b = A @ x + s
    return {"A": A, "b": b, "c": c}, float(c @ x), (x, y, s)


# ---- named workloads ----
def workload(name):
    """name -> (K, n, nnz_per_col, seed)"""
    if name == "config1_lp":      # BASELINE.json configs[0]
        return {"l": 4000}, 2000, 50, 1
    if name == "config2_lp_soc":  # configs[1]: m=2e5, n=1e5, nnz~2e6
        return {"l": 100000, "q": [10] * 10000}, 100000, 20, 2
    if name == "target_lp_soc":   # metric workload: m=2e6, n=1e6, nnz~2e7
        return {"l": 1000000, "q": [10] * 100000}, 1000000, 20, 5
    if name == "target_lp":
        return {"l": 2000000}, 1000000, 20, 5
    if name == "target_qp":       # the metric workload with a quadratic objective: P = I + B'B, nnz(triu P) ~ 1e7 (K3)
        return {"l": 1000000, "q": [10] * 100000}, 1000000, 20, 6
    if name == "powerlaw_lp":     # layout robustness: metric size, heavy-tailed row lengths (pattern: workload_pattern)
        return {"l": 2000000}, 1000000, 20, 11
    if name == "banded_lp":       # layout robustness: metric size, banded
        return {"l": 2000000}, 1000000, 20, 12
    if name == "config4_psd":     # BASELINE.json configs[3]: PSD-heavy, 50 matrices of order 200 + l
        return {"l": 1000, "s": [200] * 50}, 335000, 30, 4
    if name == "config3_mixed":   # BASELINE.json configs[2]: z / l / box (99,999 bounds) / q / ep / ed / p, m = 999,999
        rng = np.random.default_rng(3)
        return {"z": 100000, "l": 300000, "bu": rng.uniform(0.5, 2.0, 99999).tolist(), "bl": (-rng.uniform(0.5, 2.0, 99999)).tolist(),
                "q": [20] * 5000, "ep": 50000, "ed": 50000,
                "p": (rng.uniform(0.1, 0.9, 33333) * rng.choice([-1.0, 1.0], 33333)).tolist()}, 500000, 20, 3
    if name == "small_lp_soc":    # smoke / CI size
        return {"l": 2000, "q": [10] * 200}, 2000, 20, 7
    if name == "config5_small":   # one problem of the 512-problem batch
        return {"l": 2000, "q": [50] * 20, "s": [20] * 5}, 1350, 40, 1000
    raise KeyError(name)


def workload_pattern(name):
    """sparsity pattern family of a named workload (gen_feasible's `pattern`)"""
    return {"powerlaw_lp": "powerlaw", "banded_lp": "banded"}.get(name, "uniform")


def workload_qp(name):
    """None, or gen_feasible_qp's `b_per_col` for the named QP workload (rows of B = n / 4: nnz(triu P) ~ (2 b_per_col^2 + 0.5) n)"""
    return {"target_qp": 2}.get(name)


def gen_feasible_qp(K, n, nnz_per_col, seed, proj_dual, p_diag=1.0, b_per_col=3):
    """Strictly convex QP over the cone K: P = p_diag*I + B'B (sparse, PD) makes x — hence s — unique,
    and with n >= #active rows the dual y is unique too, so x, y, s can all be compared entry-wise.
    Returns (data with P upper-triangular CSC, p_star, (x, y, s))."""
    rng = np.random.default_rng(seed)
    m = cone_dims(K)
    z = rng.standard_normal(m)
    y = np.asarray(proj_dual(z, K), dtype=np.float64)
    s = y - z
    A = random_sparse(m, n, nnz_per_col, rng)
    B = random_sparse(max(n // 4, 1), n, b_per_col, rng)
    P = (B.T @ B + p_diag * sparse.eye(n)).tocsc()
    P.sort_indices()
    x = rng.standard_normal(n)
    c = -(P @ x) - (A.T @ y)
    b = A @ x + s
    Pu = sparse.triu(P, format="csc")
    Pu.sort_indices()
    return {"P": Pu, "A": A, "b": b, "c": c}, float(0.5 * x @ (P @ x) + c @ x), (x, y, s)
