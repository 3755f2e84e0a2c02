"""K9 decision lab: would a GEMM-only PSD projection — Pi_+(X) = (X + |X|) / 2 with |X| = X sign(X) from a
Newton-Schulz / scaled sign iteration, every step two order-200 GEMMs on the matrix cores — beat the block-Jacobi sweeps?
Iteration counts are a property of the spectra, so they are measured here in numpy on REAL iterates of config 4 (the matrices the
cone kernel projects at ADMM iterations 5, 50, 500: z = y - s of a solve stopped there), and priced with the measured cost of
the existing MFMA GEMM launch (k_psd_gemm: 38 us per 50 x (200 x 200 x 200), profiles/r02_psd_mfma.txt = 21 TFLOP/s).
    python tools/psd_sign_lab.py > profiles/r03_psd_sign_lab.txt      (GPU box: the iterates come from the HIP solver)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import scs
from scs import _scs_hip
import problem_gen as pg
import helpers

GEMM_US = 38.0      # one launch of k_psd_gemm for 50 matrices of order 200 (measured, r02)
JACOBI_MS = 1.8     # current K9 per projection inside the config-4 solve (r02: sweeps 1.07 + apply_v 0.24 + GEMMs 0.17 + ...)
TOL = 1e-10

proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config4_psd")
data, _, _ = pg.gen_feasible(K, n, k, seed, proj)
o, d, order = K["l"], 200 * 201 // 2, 200


def ns_iters(X, exact, scaled):
    """Newton-Schulz iterations until ||Pi_+ - exact||_F <= TOL ||X||_F.  scaled: start from X / ||X||_2 (a power-iteration
    estimate costs GEMV only) instead of X / ||X||_F"""
    nx = np.linalg.norm(X, 2) * 1.0001 if scaled else np.linalg.norm(X)
    Y = X / nx
    I = np.eye(X.shape[0])
    for it in range(1, 200):
        Y = 0.5 * Y @ (3 * I - Y @ Y)
        P = 0.5 * (X + X @ Y)
        if np.linalg.norm(P - exact) <= TOL * np.linalg.norm(X):
            return it
    return 200


def newton_inv_iters(X, exact):
    """scaled Newton with inverses (Y <- (mu Y + (mu Y)^-1) / 2, determinantal-free norm scaling): what a QDWH-like method
    with a dense solve per step would need"""
    Y = X / np.linalg.norm(X, 2)
    for it in range(1, 60):
        Yi = np.linalg.inv(Y)
        mu = np.sqrt(np.linalg.norm(Yi, 1) * np.linalg.norm(Yi, np.inf) / (np.linalg.norm(Y, 1) * np.linalg.norm(Y, np.inf))) ** 0.5
        Y = 0.5 * (mu * Y + Yi / mu)
        P = 0.5 * (X + X @ Y)
        if np.linalg.norm(P - exact) <= TOL * np.linalg.norm(X):
            return it
    return 60


print("# config 4 (50 PSD cones of order 200): spectra of the projected matrices and sign-iteration counts to %.0e relative accuracy" % TOL)
for iters in (5, 50, 500):
    sol = scs.SCS(data, K, verbose=False, max_iters=iters, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0).solve(warm_start=False)
    z = sol["y"] - sol["s"]
    rel_min, ns, nss, nw = [], [], [], []
    for i in range(0, 50, 5):  # every fifth block
        X = helpers.svec_to_sym(z[o + i * d:o + (i + 1) * d], order)
        w, U = np.linalg.eigh(X)
        exact = (U * np.maximum(w, 0.0)) @ U.T
        big = np.abs(w) > TOL * np.abs(w).max()
        rel_min.append(np.abs(w[big]).min() / np.abs(w).max())
        ns.append(ns_iters(X, exact, False))
        nss.append(ns_iters(X, exact, True))
        nw.append(newton_inv_iters(X, exact))
    gemm_ms = 2 * max(nss) * GEMM_US * 1e-3 + 2 * GEMM_US * 1e-3
    print("ADMM iteration %3d: smallest relevant |lambda| / |lambda|max over the blocks: median %.1e, min %.1e" % (iters, np.median(rel_min), np.min(rel_min)))
    print("   Newton-Schulz (X / ||X||_F start):   iterations min / median / max = %d / %d / %d" % (min(ns), int(np.median(ns)), max(ns)))
    print("   Newton-Schulz (X / ||X||_2 start):   iterations min / median / max = %d / %d / %d  -> %d GEMM launches = %.2f ms per projection (Jacobi now: %.1f ms)"
          % (min(nss), int(np.median(nss)), max(nss), 2 * max(nss) + 2, gemm_ms, JACOBI_MS))
    print("   scaled Newton with a dense inverse per step: iterations min / median / max = %d / %d / %d" % (min(nw), int(np.median(nw)), max(nw)))
