#!/usr/bin/env python3
"""numpy model of the two-sided block Jacobi of psd.hpp: how many outer sweeps does a pivot solve of
(a) one full cyclic sweep over the 16x16 pivot (15 rounds, what wave_jacobi16 does) need, versus
(b) only the 64 cross pairs (i in block p, j in block q; 8 rounds) with the intra-block pairs swept once per outer sweep?
Stop: ||offdiag||_F <= 1e-8 ||A||_F (kPsdOffTol2 = 1e-16)."""
import sys
import numpy as np

B = 8


def rr_pair(r, k, N):
    if k == 0:
        p, q = N - 1, r % (N - 1)
    else:
        p, q = (r + k) % (N - 1), (r - k + (N - 1)) % (N - 1)
    return (p, q) if p < q else (q, p)


def rot(S, W, i, j):
    apq = S[i, j]
    if abs(apq) < 1e-300:
        return
    theta = (S[j, j] - S[i, i]) / (2 * apq)
    t = np.sign(theta) / (abs(theta) + np.sqrt(theta * theta + 1)) if theta != 0 else 1.0
    c = 1 / np.sqrt(1 + t * t)
    s = t * c
    J = np.eye(S.shape[0])
    J[i, i] = c; J[j, j] = c; J[i, j] = s; J[j, i] = -s
    S[:] = J.T @ S @ J
    W[:] = W @ J


def pivot_full(S):
    W = np.eye(16)
    for r in range(15):
        for k in range(8):
            p, q = rr_pair(r, k, 16)
            rot(S, W, p, q)
    return W


def pivot_cross(S, intra):
    W = np.eye(16)
    if intra:  # the two 8x8 diagonal blocks: one cyclic sweep each (7 rounds of 4 + 4 rotations)
        for r in range(7):
            for k in range(4):
                p, q = rr_pair(r, k, 8)
                rot(S, W, p, q)
                rot(S, W, 8 + p, 8 + q)
    for r in range(8):
        for i in range(8):
            rot(S, W, i, 8 + (i + r) % 8)
    return W


def sweeps(A, mode, max_sweeps=30):
    n = A.shape[0]
    NB = n // B
    A = A.copy()
    hist = []
    for sweep in range(max_sweeps):
        off = np.sqrt(max(np.sum(A * A) - np.sum(np.diag(A) ** 2), 0)); tot = np.linalg.norm(A)
        hist.append(off / tot)
        if off <= 1e-8 * tot:
            return sweep, hist
        done_intra = set()
        for r in range(NB - 1):
            for k in range(NB // 2):
                p, q = rr_pair(r, k, NB)
                idx = np.r_[p * B:(p + 1) * B, q * B:(q + 1) * B]
                S = A[np.ix_(idx, idx)].copy()
                if mode == "full":
                    W = pivot_full(S)
                else:
                    intra = r == 0 if mode == "cross" else False   # "cross": every block is in exactly one pivot of step 0
                    W = pivot_cross(S, intra)
                A[:, idx] = A[:, idx] @ W
                A[idx, :] = W.T @ A[idx, :]
    return max_sweeps, hist


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    rng = np.random.RandomState(0)
    M = rng.randn(n, n); M = (M + M.T) / 2
    w, V = np.linalg.eigh(M)
    for label, A in (("cold (random symmetric)", M),
                     ("warm 1e-2", V.T @ (M + 1e-2 * (lambda E: (E + E.T) / 2)(rng.randn(n, n))) @ V),
                     ("warm 1e-4", V.T @ (M + 1e-4 * (lambda E: (E + E.T) / 2)(rng.randn(n, n))) @ V)):
        for mode in ("full", "cross", "cross_nointra"):
            ns, hist = sweeps(A, mode)
            print("%-26s %-14s sweeps %2d   off/tot per sweep: %s" % (label, mode, ns, " ".join("%.1e" % h for h in hist)))
