"""Round 5 prototype (numpy) of K9's GEMM-only fast path on the matrices of tools/dbg/psd_dump_iterates.py:
one Ogita-Aishima-style step on the MIXED-SIGN pairs of S = V'AV (V = the previous call's eigenbasis),
    K_ij = S_ij / (d_j - d_i)  (d_i d_j < 0),   Q = I + K + K^2/2 (+ K^3/6),   V1 = V Q,   S1 = V1' A V1,
then the existing second-order reconstruction X+ = V1 F(S1) V1'.  Reports what the device would test (off-norms) and the
error against LAPACK.
    python tools/dbg/psd_refine_proto.py gpurun_out/psd_iterates.npz"""
import sys
import numpy as np

ORDER = 200


def svec_to_sym(v, n=ORDER):
    X = np.zeros((n, n))
    idx = 0
    for j in range(n):
        X[j:, j] = v[idx:idx + n - j]
        idx += n - j
    X[np.triu_indices(n, 1)] = 0
    d = np.diag(X).copy()
    X = X / np.sqrt(2.0)
    X = X + X.T
    X[np.diag_indices(n)] = d
    return X


def proj_exact(A):
    w, U = np.linalg.eigh(A)
    return (U * np.maximum(w, 0)) @ U.T


def dk_map(S):
    d = np.diag(S).copy()
    hi = np.maximum.outer(d, d)
    lo = np.minimum.outer(d, d)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.where(lo > 0, 1.0, np.where(hi <= 0, 0.0, hi / (hi - lo)))
    F = S * g
    F[np.diag_indices_from(F)] = np.maximum(d, 0)
    return F


def offs(S):
    d = np.diag(S)
    sg = d > 0
    mixed = sg[:, None] != sg[None, :]
    off = S - np.diag(d)
    return np.linalg.norm(off), np.linalg.norm(off[mixed]), np.linalg.norm(S)


def oa_step(S, order=2, allpairs=False, cap=None):
    d = np.diag(S)
    sg = d > 0
    mixed = sg[:, None] != sg[None, :]
    den = d[None, :] - d[:, None]  # d_j - d_i
    with np.errstate(divide="ignore", invalid="ignore"):
        K = np.where(mixed, S / den, 0.0)
        if allpairs:
            Ka = np.where(~np.eye(len(d), dtype=bool), S / den, 0.0)
            ok = np.abs(Ka) < (cap if cap else 0.05)
            K = np.where(mixed, K, np.where(ok, Ka, 0.0))
    K = np.nan_to_num(K)
    Q = np.eye(len(d)) + K + 0.5 * K @ K
    if order >= 3:
        Q += K @ K @ K / 6.0
    return K, Q


def main(path):
    z = np.load(path)
    blocks = z["blocks"]
    for k0 in (100, 300, 600):
        print("=== iterations %d.. ===" % k0)
        for bi in range(len(blocks)):
            A0 = svec_to_sym(z["z_%d" % k0][bi])
            w, V = np.linalg.eigh(A0)
            nrm = np.linalg.norm(A0)
            wpos = w[w > 0].min() if (w > 0).any() else np.nan
            wneg = -w[w <= 0].max() if (w <= 0).any() else np.nan
            print("block %2d: |A|_F %.3e  lam_max %.3e  min pos %.2e  min |neg| %.2e (rel to |A|_F: %.1e %.1e)  npos %d" % (
                blocks[bi], nrm, np.abs(w).max(), wpos, wneg, wpos / nrm, wneg / nrm, (w > 0).sum()))
            Vc = V.copy()  # chained: never re-diagonalised exactly again
            for it in range(k0 + 1, k0 + 12):
                A = svec_to_sym(z["z_%d" % it][bi])
                ex = proj_exact(A)
                S = Vc.T @ A @ Vc
                S = 0.5 * (S + S.T)
                o_all, o_mix, tot = offs(S)
                K, Q = oa_step(S, order=3)
                V1 = Vc @ Q
                S1 = V1.T @ A @ V1
                S1 = 0.5 * (S1 + S1.T)
                o1_all, o1_mix, _ = offs(S1)
                X1 = V1 @ dk_map(S1) @ V1.T
                err1 = np.linalg.norm(X1 - ex) / tot
                # plain DK on S without any step (what the current code would do if it accepted S as is)
                err0 = np.linalg.norm(Vc @ dk_map(S) @ Vc.T - ex) / tot
                # second step
                K2, Q2 = oa_step(S1, order=3)
                V2 = V1 @ Q2
                S2 = V2.T @ A @ V2
                S2 = 0.5 * (S2 + S2.T)
                o2_all, o2_mix, _ = offs(S2)
                err2 = np.linalg.norm(V2 @ dk_map(S2) @ V2.T - ex) / tot
                orth = np.linalg.norm(V1.T @ V1 - np.eye(ORDER))
                print("   it %d: off(S) all %.1e mixed %.1e | max|K| %.1e |K|_F %.1e | after 1 step: all %.1e mixed %.1e err %.1e (no step: %.1e) orth %.1e | 2 steps: mixed %.1e err %.1e" % (
                    it, o_all / tot, o_mix / tot, np.abs(K).max(), np.linalg.norm(K), o1_all / tot, o1_mix / tot, err1, err0, orth, o2_mix / tot, err2))
                Vc = V1  # chained basis (one step per call)


if __name__ == "__main__":
    main(sys.argv[1])
