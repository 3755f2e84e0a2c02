"""CPU prototype (numpy): config 3's reduced KKT system G = rho_x I + A' W A with the zero-cone rows weighted 1000 x.
Compares Jacobi-PCG (what ships, what the reference's indirect backend does) with a Woodbury preconditioner that takes
the zero-cone block exactly:  M = D + w_z Az' Az,  M^-1 = D^-1 - D^-1 Az' S^-1 Az D^-1,  S = I / w_z + Az D^-1 Az',
S^-1 applied by k inner Jacobi-CG steps (or exactly).  1/10 of config 3's size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
from scipy import sparse
from scipy.sparse import linalg as sla
import problem_gen as pg
from oracle import scs_oracle

sc = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(3)
nb = 99999 // sc
K = {"z": 100000 // sc, "l": 300000 // sc, "bu": rng.uniform(0.5, 2.0, nb).tolist(), "bl": (-rng.uniform(0.5, 2.0, nb)).tolist(),
     "q": [20] * (5000 // sc), "ep": 50000 // sc, "ed": 50000 // sc,
     "p": (rng.uniform(0.1, 0.9, 33333 // sc) * rng.choice([-1.0, 1.0], 33333 // sc)).tolist()}
n = 500000 // sc
data, p_star, _ = pg.gen_feasible(K, n, 20, 3, lambda z, K: scs_oracle.proj_cone(z, K, dual=True))
A = data["A"].copy()
m = A.shape[0]
Ax = scs_oracle.normalize(A, None, data["b"], data["c"], K)[0]
A = sparse.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
scale, rho_x = 0.1, 1e-6
z = K["z"]
w = np.full(m, scale); w[:z] = 1000.0 * scale        # R_y^-1
Ar = A.tocsr()
Az = Ar[:z]
Al = Ar[z:]
G = lambda x: rho_x * x + A.T @ (w * (A @ x))
dG = rho_x + (A.multiply(A)).T @ w                     # Jacobi of the whole
dGl = rho_x + (Al.multiply(Al)).T @ w[z:]              # Jacobi of the part without the zero-cone rows
wz = 1000.0 * scale
rhs = rng.standard_normal(n)

def pcg(apply_M, tol=1e-9, maxit=5000, flexible=False):
    x = np.zeros(n); r = rhs.copy(); zv = apply_M(r); p = zv.copy(); rz = r @ zv; it = 0
    nb = np.abs(rhs).max()
    while np.abs(r).max() > tol * nb and it < maxit:
        Gp = G(p); a = rz / (p @ Gp); x += a * p; r_old = r.copy(); r -= a * Gp
        z_old = zv; zv = apply_M(r)
        rz_new = r @ zv
        beta = (rz_new - (r_old @ zv if flexible else 0.0)) / rz if flexible else rz_new / rz
        rz = rz_new; p = zv + beta * p; it += 1
    return it, np.abs(rhs - G(x)).max() / nb

print("n=%d m=%d z=%d nnz=%d" % (n, m, z, A.nnz))
print("Jacobi (whole diag):            steps %d  res %.1e" % pcg(lambda r: r / dG))
Dinv = 1.0 / dGl
Sdiag = 1.0 / wz + (Az.multiply(Az)) @ Dinv
Smat = (sparse.identity(z) / wz + Az @ sparse.diags(Dinv) @ Az.T).tocsc()
ev = sla.eigsh(sparse.diags(Sdiag ** -0.5) @ Smat @ sparse.diags(Sdiag ** -0.5), k=1, which="LA", return_eigenvectors=False)
ev2 = sla.eigsh(sparse.diags(Sdiag ** -0.5) @ Smat @ sparse.diags(Sdiag ** -0.5), k=1, which="SA", return_eigenvectors=False)
print("Jacobi-scaled S spectrum: [%.3f, %.3f]" % (ev2[0], ev[0]))
lu = sla.splu(Smat)
def M_exact(r):
    t = Dinv * r
    return t - Dinv * (Az.T @ lu.solve(Az @ t))
print("Woodbury, S^-1 exact:           steps %d  res %.1e" % pcg(M_exact))
inner_total = [0]
def make_inner(k, itol=0.0):
    def inner(bv):
        y = np.zeros(z); r = bv.copy(); zz = r / Sdiag; p = zz.copy(); rz = r @ zz
        for _ in range(k):
            Sp = Smat @ p; a = rz / (p @ Sp); y += a * p; r -= a * Sp
            inner_total[0] += 1
            if itol > 0 and np.abs(r).max() <= itol * np.abs(bv).max(): break
            zz = r / Sdiag; rz2 = r @ zz; p = zz + (rz2 / rz) * p; rz = rz2
        return y
    return inner
for k in (2, 4, 6, 8, 12):
    inner = make_inner(k)
    def M_k(r):
        t = Dinv * r
        return t - Dinv * (Az.T @ inner(Az @ t))
    for flex in (False, True):
        it, res = pcg(M_k, flexible=flex)
        print("Woodbury, %2d inner CG steps %s: outer steps %d  res %.1e" % (k, "(flexible)" if flex else "          ", it, res))
# Chebyshev inner (fixed polynomial => a fixed linear preconditioner), bounds from the measured spectrum widened by 10 %
lo, hi = ev2[0] * 0.9, ev[0] * 1.1
def make_cheb(k):
    th, de = (hi + lo) / 2, (hi - lo) / 2
    def inner(bv):   # Chebyshev iteration on Ds^-1 S
        y = np.zeros(z); r = bv.copy(); sg1 = th / de; rho = 1 / sg1; d = (r / Sdiag) / th
        for i in range(k):
            y += d; r -= Smat @ d
            rho_n = 1 / (2 * sg1 - rho); d = rho_n * rho * d + 2 * rho_n / de * (r / Sdiag); rho = rho_n
        return y
    return inner
for k in (3, 4, 5, 6, 8):
    inner = make_cheb(k)
    def M_c(r):
        t = Dinv * r
        return t - Dinv * (Az.T @ inner(Az @ t))
    it, res = pcg(M_c)
    print("Woodbury, Chebyshev-%d inner:      outer steps %d  res %.1e" % (k, it, res))
