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
