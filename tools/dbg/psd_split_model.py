"""Round 6 model (numpy) of K9's flow with SIGN-SPLIT sweeps on the matrices config 4 really projects
(tools/dbg/psd_dump_iterates2.py): the warm basis V is carried from call to call exactly as the device does, S = V'AV is ordered by the
sign of its diagonal (positives | padding | negatives, 8-blocks), and a projection that is neither converged nor refinable runs

  flow "full"  (round 5): block-Jacobi sweeps over ALL block pairs (25 outer steps at order 200) until converged or refinable,
  flow "split" (round 6): while both diagonal blocks are safely definite (omega <= OMEGA_SPLIT) only the pivots between a positive and a
                          negative 8-block — a bipartite tournament of max(a, NB - a) outer steps; the blocks idle in a step are paired
                          among themselves (same-sign rotations for free) — until the MIXED-sign part is below tol (the reconstruction
                          is then exact whatever is left inside the two blocks) or the matrix is refinable; full sweeps otherwise,

followed by the GEMM-only refinement where the gate allows it.  Prints per phase of the solve: outer steps per projection (the latency
chain of the sweep kernel: ~13 us each, and the V update is proportional to it), how the calls ended, and the error against LAPACK.
    python tools/dbg/psd_split_model.py gpurun_out/psd_iterates2.npz [omega_split]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from psd_refine_proto import svec_to_sym, proj_exact, dk_map, ORDER

B = 8
TOL = 1e-8
K_GATE, OFF_GATE, OM_GATE, OM_RELAXED = 6e-3, 1.5e-3, 0.05, 0.25
OMEGA_SPLIT = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25


def rr_pair(r, k, N):
    if k == 0:
        p, q = N - 1, r % (N - 1)
    else:
        p, q = (r + k) % (N - 1), (r - k + (N - 1)) % (N - 1)
    return (p, q) if p < q else (q, p)


def stats(S, real):
    """gate statistics over the REAL positions (`real`: boolean mask; the padding moves with the sign sort)"""
    T = S[np.ix_(real, real)]
    n = T.shape[0]
    d = np.diag(T)
    sg = d > 0
    E = T - np.diag(d)
    mixed = sg[:, None] != sg[None, :]
    same = ~mixed & ~np.eye(n, dtype=bool)
    tot = np.linalg.norm(T)
    den = d[None, :] - d[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        K1 = np.where(mixed, E / den, 0.0)
        om = np.where(same & (E != 0), E * E / np.abs(np.outer(d, d)), 0.0).sum()
    return dict(off=np.linalg.norm(E) / tot, mix=np.linalg.norm(E[mixed]) / tot, kf=np.linalg.norm(K1), om=om, K1=K1, npos=int(sg.sum()))


def pivot_cross(S, intra):
    """the rotations wave_jacobi16 applies to a 16x16 pivot: (intra: the 2 x 28 pairs inside the diagonal blocks,) the 64 cross pairs"""
    W = np.eye(16)

    def rot(i, j):
        apq = S[i, j]
        if abs(apq) < 1e-300:
            return
        theta = (S[j, j] - S[i, i]) / (2 * apq)
        t = np.sign(theta) / (abs(theta) + np.sqrt(theta * theta + 1)) if theta != 0 else 1.0
        c = 1 / np.sqrt(1 + t * t)
        s = t * c
        ci, cj = S[:, i].copy(), S[:, j].copy()
        S[:, i], S[:, j] = c * ci - s * cj, s * ci + c * cj
        ri, rj = S[i, :].copy(), S[j, :].copy()
        S[i, :], S[j, :] = c * ri - s * rj, s * ri + c * rj
        wi, wj = W[:, i].copy(), W[:, j].copy()
        W[:, i], W[:, j] = c * wi - s * wj, s * wi + c * wj
    if intra:
        for r in range(7):
            for k in range(4):
                p, q = rr_pair(r, k, 8)
                rot(p, q)
                rot(8 + p, 8 + q)
    for r in range(8):
        for i in range(8):
            rot(i, 8 + (i + r) % 8)
    return W


def schedule_full(NB):
    return [[rr_pair(r, k, NB) for k in range(NB // 2)] for r in range(NB - 1)]


def schedule_split(NB, a):
    """bipartite tournament between the blocks [0, a) and [a, NB): step r pairs block i of the smaller side with block (i + r) mod L of
    the larger one; the L - s blocks of the larger side that are idle in a step are paired among themselves"""
    small, large = (list(range(a)), list(range(a, NB))) if a <= NB - a else (list(range(a, NB)), list(range(a)))
    s, L = len(small), len(large)
    steps = []
    for r in range(L):
        pairs, used = [], set()
        for i in range(s):
            j = large[(i + r) % L]
            used.add(j)
            pairs.append((min(small[i], j), max(small[i], j)))
        idle = [j for j in large if j not in used]
        for t in range(0, len(idle) - 1, 2):
            pairs.append((idle[t], idle[t + 1]))
        steps.append(pairs)
    return steps


def sweep(S, V, steps, intra_first):
    for r, pairs in enumerate(steps):
        for (p, q) in pairs:
            idx = np.r_[p * B:(p + 1) * B, q * B:(q + 1) * B]
            W = pivot_cross(S[np.ix_(idx, idx)].copy(), intra_first and r == 0)
            S[:, idx] = S[:, idx] @ W
            S[idx, :] = W.T @ S[idx, :]
            V[:, idx] = V[:, idx] @ W
    S[:] = 0.5 * (S + S.T)
    return len(steps)


def refine(S, real, K1):
    T = S[np.ix_(real, real)]
    d = np.diag(T)
    E = T - np.diag(d)
    sg = d > 0
    mixed = sg[:, None] != sg[None, :]
    den = d[None, :] - d[:, None]
    EK = E @ K1
    with np.errstate(divide="ignore", invalid="ignore"):
        K2 = np.where(mixed, (E + EK + EK.T) / den, 0.0)
    Q = np.eye(S.shape[0])
    Q[np.ix_(real, real)] += K2 + 0.5 * K2 @ K2
    return Q


def sort_by_sign(S, V, real):
    """positives | padding | negatives; returns the number of 8-blocks on the positive side (its last block padded), or None when the
    padding does not reach the next block boundary.  `real` (which positions are not padding) is permuted along."""
    NP = S.shape[0]
    d = np.diag(S)
    pos = [i for i in range(NP) if real[i] and d[i] > 0]
    neg = [i for i in range(NP) if real[i] and not d[i] > 0]
    pad = [i for i in range(NP) if not real[i]]
    need = (-len(pos)) % B
    if need > len(pad):
        return None
    perm = pos + pad[:need] + neg + pad[need:]
    # (the padding behind the negatives: whole blocks of zeros or a tail — never rotated, a_pq = 0)
    S[:] = S[np.ix_(perm, perm)]
    V[:] = V[:, perm]
    real[:] = real[perm]
    return (len(pos) + need) // B


def project(A, V, flow, n, real):
    """one call of the pipeline; V (NP x NP, padded) and `real` are updated in place.  Returns (X+, outer steps, how it ended)"""
    NP = V.shape[0]
    NB = NP // B
    Ap = np.zeros((NP, NP))
    Ap[:n, :n] = A
    S = V.T @ Ap @ V
    S = 0.5 * (S + S.T)
    steps, how = 0, ""
    a = sort_by_sign(S, V, real) if flow == "split" else None
    for trial in range(12):
        st = stats(S, real)
        if st["off"] <= TOL or (st["mix"] <= TOL and st["om"] <= OM_RELAXED and how != ""):
            how += "C"
            break
        if flow == "split" and st["mix"] <= TOL and st["om"] <= OM_RELAXED:
            how += "C"  # arrives split (and definite): nothing to do
            break
        if st["kf"] <= K_GATE and st["off"] <= OFF_GATE and st["om"] <= OM_GATE and "R" not in how:
            Q = refine(S, real, st["K1"])
            S = Q.T @ S @ Q
            S = 0.5 * (S + S.T)
            V[:] = V @ Q
            how += "R"
            continue
        if flow == "split" and a is not None and st["om"] <= OMEGA_SPLIT and 0 < a < NB:
            # the signs of the diagonal may have changed since the sort (few do): the schedule follows the POSITIONS, the tests the signs
            steps += sweep(S, V, schedule_split(NB, a), False)
            how += "s"
        else:
            steps += sweep(S, V, schedule_full(NB), True)
            how += "F"
    # reconstruction: second-order map of (D + E) in the (possibly permuted) basis; the padding carries zeros
    X = (V @ dk_map(S.copy()) @ V.T)[:n, :n]
    return X, steps, how


def main(path):
    z = np.load(path)
    blocks, starts = z["blocks"], z["starts"]
    n = ORDER
    NP = ((n + B - 1) // B + 1) // 2 * 2 * B
    print("omega_split = %g; order %d padded to %d (%d blocks)" % (OMEGA_SPLIT, n, NP, NP // B))
    print("%5s | %-28s | %-28s" % ("iter", "flow full (round 5)", "flow split (round 6)"))
    tot = {"full": 0, "split": 0}
    for k0 in starts:
        rows = {}
        for flow in ("full", "split"):
            rec = []
            for bi in range(len(blocks)):
                A0 = svec_to_sym(z["z_%d" % k0][bi])
                w, U = np.linalg.eigh(A0)
                V = np.eye(NP)
                V[:n, :n] = U
                real = np.arange(NP) < n
                for it in range(k0 + 1, k0 + 4):
                    A = svec_to_sym(z["z_%d" % it][bi])
                    X, steps, how = project(A, V, flow, n, real)
                    err = np.linalg.norm(X - proj_exact(A)) / np.linalg.norm(A)
                    orth = np.linalg.norm(V.T @ V - np.eye(NP))
                    rec.append((it, bi, steps, how, err, orth))
            rows[flow] = rec
        for it in range(k0 + 1, k0 + 4):
            cells = []
            for flow in ("full", "split"):
                rr = [r for r in rows[flow] if r[0] == it]
                st = np.mean([r[2] for r in rr])
                tot[flow] += sum(r[2] for r in rr)
                hows = {}
                for r in rr:
                    hows[r[3]] = hows.get(r[3], 0) + 1
                cells.append("%5.1f steps %-12s err %.0e" % (st, ",".join("%s:%d" % kv for kv in sorted(hows.items())), max(r[4] for r in rr)))
            print("%5d | %-28s | %-28s" % (it, cells[0], cells[1]))
    print("total outer steps: full %d, split %d (%.2f x)" % (tot["full"], tot["split"], tot["full"] / max(tot["split"], 1)))
    # what the position of the sign boundary looks like
    for k0 in starts:
        A0 = svec_to_sym(z["z_%d" % k0][0])
        w = np.linalg.eigvalsh(A0)
        print("iter %d block 0: %d positive of %d eigenvalues, smallest |lambda| / |A| = %.1e" % (k0, (w > 0).sum(), n, np.abs(w).min() / np.linalg.norm(A0)))


if __name__ == "__main__":
    main(sys.argv[1])
