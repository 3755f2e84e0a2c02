"""Variants of the refinement step (see psd_refine_proto.py), chained over consecutive iterations:
  a  mixed pairs, first order          b  all pairs (|K_ij| <= cap), first order
  c  mixed pairs, second order K2 = K1 - off([E,K1])/2 ./ den      d  all pairs, second order
Q = I + K + K^2/2 + K^3/6.  Reports mixed off-norm of S1 (what the device tests), omega = sum_same E_ij^2/(d_i d_j), error vs LAPACK."""
import sys
import numpy as np
from psd_refine_proto import svec_to_sym, proj_exact, dk_map, offs, ORDER

CAP = float(__import__("os").environ.get("CAP", "0.1"))


def gen_K(S, allpairs, second):
    n = len(S)
    d = np.diag(S).copy()
    E = S - np.diag(d)
    sg = d > 0
    mixed = sg[:, None] != sg[None, :]
    den = d[None, :] - d[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(den != 0, 1.0 / den, 0.0)
    def mask(K):
        if allpairs:
            sel = mixed | (np.abs(K) <= CAP)
        else:
            sel = mixed
        return np.where(sel, K, 0.0)
    K1 = mask(E * inv)
    if not second:
        return K1
    EK = E @ K1
    C = EK + EK.T  # [E, K1]
    C -= np.diag(np.diag(C))
    K2 = mask((E + 0.5 * C) * inv)
    return K2


def omega(S):
    d = np.diag(S)
    sg = d > 0
    same = (sg[:, None] == sg[None, :]) & ~np.eye(len(d), dtype=bool)
    E2 = S * S
    dd = np.abs(np.outer(d, d))
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(same & (E2 > 0), E2 / dd, 0.0)
    return t.sum()


def main(path):
    z = np.load(path)
    blocks = z["blocks"]
    for k0 in (100, 300, 600):
        print("=== iterations %d.. ===" % k0)
        for bi in range(0, len(blocks), 2):
            A0 = svec_to_sym(z["z_%d" % k0][bi])
            w, V = np.linalg.eigh(A0)
            for name, allp, sec in (("a", False, False), ("b", True, False), ("c", False, True), ("d", True, True)):
                Vc = V.copy()
                line = []
                for it in range(k0 + 1, k0 + 11):
                    A = svec_to_sym(z["z_%d" % it][bi])
                    ex = proj_exact(A)
                    S = Vc.T @ A @ Vc
                    S = 0.5 * (S + S.T)
                    o_all, o_mix, tot = offs(S)
                    K = gen_K(S, allp, sec)
                    K2m = K @ K
                    Q = np.eye(ORDER) + K + 0.5 * K2m + K2m @ K / 6.0
                    V1 = Vc @ Q
                    S1 = V1.T @ A @ V1
                    S1 = 0.5 * (S1 + S1.T)
                    o1_all, o1_mix, _ = offs(S1)
                    err1 = np.linalg.norm(V1 @ dk_map(S1) @ V1.T - ex) / tot
                    orth = np.linalg.norm(V1.T @ V1 - np.eye(ORDER))
                    line.append("%d: in %.0e/%.0e K %.0e -> mix %.1e all %.0e om %.0e err %.0e orth %.0e" % (
                        it, o_all / tot, o_mix / tot, np.abs(K).max(), o1_mix / tot, o1_all / tot, omega(S1), err1, orth))
                    Vc = V1
                print("block %d variant %s" % (blocks[bi], name))
                for l in (line[0], line[1], line[4], line[9]):
                    print("    " + l)


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main(sys.argv[1])
