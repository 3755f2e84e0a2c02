"""Model (numpy) of the round-5 K9 flow on real consecutive config-4 iterates (tools/dbg/psd_dump_iterates.py):
  S = V'AV -> [Jacobi sweeps until converged (all-off <= tol) or REFINABLE (gate)] -> one second-order refinement of the sign split
      K1 = mixed(S ./ den),  K2 = mixed((S_off + [S_off, K1]) ./ den),  Q = I + K2 + K2^2/2,  V <- V Q,  S1 = Q' S Q
  -> relaxed test on S1 (mixed-off <= tol |S|, omega <= omega_max) else more sweeps -> X+ = V F(S1) V'.
A 'sweep' is modelled by an exact re-diagonalisation (LAPACK).  Prints how often each path is taken and the errors."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from psd_refine_proto import svec_to_sym, proj_exact, dk_map, offs, ORDER
from psd_refine_proto2 import omega

KAPPA = float(sys.argv[2]) if len(sys.argv) > 2 else 6e-3
OFFMAX = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5e-3
TOL = 1e-8


def gate(S):
    d = np.diag(S); E = S - np.diag(d)
    sg = d > 0; mixed = sg[:, None] != sg[None, :]
    den = d[None, :] - d[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        K1 = np.where(mixed, E / den, 0.0)
    o_all, o_mix, tot = offs(S)
    return K1, np.linalg.norm(K1), o_all / tot, o_mix / tot, omega(S)


def refine(S, K1):
    d = np.diag(S); E = S - np.diag(d)
    sg = d > 0; mixed = sg[:, None] != sg[None, :]
    den = d[None, :] - d[:, None]
    EK = E @ K1
    with np.errstate(divide="ignore", invalid="ignore"):
        K2 = np.where(mixed, (E + EK + EK.T) / den, 0.0)
    Q = np.eye(len(d)) + K2 + 0.5 * K2 @ K2
    S1 = Q.T @ S @ Q
    return Q, 0.5 * (S1 + S1.T)


z = np.load(sys.argv[1]); blocks = z["blocks"]
for k0 in (100, 300, 600):
    n_fast = n_jac = n_fail = 0
    worst = 0.0; worst_mix = 0.0
    for bi in range(len(blocks)):
        w, V = np.linalg.eigh(svec_to_sym(z["z_%d" % k0][bi]))
        for it in range(k0 + 1, k0 + 12):
            A = svec_to_sym(z["z_%d" % it][bi]); ex = proj_exact(A)
            S = V.T @ A @ V; S = 0.5 * (S + S.T)
            K1, kf, oa, om_, om = gate(S)
            if oa <= TOL:
                S1 = S
            elif kf <= KAPPA and oa <= OFFMAX and om <= 0.05:
                Q, S1 = refine(S, K1)
                V = V @ Q
                _, _, oa1, om1, omg1 = gate(S1)
                worst_mix = max(worst_mix, om1)
                if om1 <= TOL and omg1 <= 0.25:
                    n_fast += 1
                else:
                    n_fail += 1
                    w, V = np.linalg.eigh(A); S1 = np.diag(w)
            else:
                n_jac += 1
                w, V = np.linalg.eigh(A); S1 = np.diag(w)
            err = np.linalg.norm(V @ dk_map(S1) @ V.T - ex) / np.linalg.norm(A)
            worst = max(worst, err)
    print("iterations %d..%d: fast %d, Jacobi %d, fast-then-failed %d; worst mixed-off after a refinement %.1e; worst error vs LAPACK %.1e; final orth %.1e" % (
        k0 + 1, k0 + 11, n_fast, n_jac, n_fail, worst_mix, worst, np.linalg.norm(V.T @ V - np.eye(ORDER))))
