"""CPU prototype: MINRES on the KKT system with the zero-cone block left un-eliminated,
[[G_l, Az'], [Az, -I/w_z]] with block-diagonal preconditioner diag(diag(G_l), diag(I/w_z + Az D^-1 Az')), vs Jacobi-CG on the reduced system."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
from scipy import sparse
from scipy.sparse import linalg as sla
import problem_gen as pg
from oracle import scs_oracle
sc = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-7
tol2 = float(sys.argv[3]) if len(sys.argv) > 3 else tol
rng = np.random.default_rng(3)
nb = 99999 // sc
K = {"z": 100000 // sc, "l": 300000 // sc, "bu": rng.uniform(0.5, 2.0, nb).tolist(), "bl": (-rng.uniform(0.5, 2.0, nb)).tolist(),
     "q": [20] * (5000 // sc), "ep": 50000 // sc, "ed": 50000 // sc,
     "p": (rng.uniform(0.1, 0.9, 33333 // sc) * rng.choice([-1.0, 1.0], 33333 // sc)).tolist()}
n = 500000 // sc
data, p_star, _ = pg.gen_feasible(K, n, 20, 3, lambda z, K: scs_oracle.proj_cone(z, K, dual=True))
A = data["A"].copy(); m = A.shape[0]
Ax = scs_oracle.normalize(A, None, data["b"], data["c"], K)[0]
A = sparse.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
scale, rho_x = 0.1, 1e-6
z = K["z"]; wz = 1000.0 * scale
w = np.full(m, scale); w[:z] = wz
Ar = A.tocsr(); Az = Ar[:z]; Al = Ar[z:]
rhs = rng.standard_normal(n)
G = lambda x: rho_x * x + A.T @ (w * (A @ x))
dG = rho_x + (A.multiply(A)).T @ w
cnt = [0]
def Gop(x): cnt[0] += 1; return G(x)
x, info = sla.cg(sla.LinearOperator((n, n), matvec=Gop), rhs, rtol=tol, maxiter=5000, M=sla.LinearOperator((n, n), matvec=lambda r: r / dG))
print("Jacobi-CG reduced: matvecs %d  relres %.1e" % (cnt[0], np.linalg.norm(rhs - G(x)) / np.linalg.norm(rhs)))
dGl = rho_x + (Al.multiply(Al)).T @ w[z:]
Gl = lambda x: rho_x * x + Al.T @ (scale * (Al @ x))
def Kop(v):
    cnt[0] += 1
    x, y = v[:n], v[n:]
    return np.concatenate([Gl(x) + Az.T @ y, Az @ x - y / wz])
Sd = 1.0 / wz + (Az.multiply(Az)) @ (1.0 / dGl)
Pinv = np.concatenate([1.0 / dGl, 1.0 / Sd])
cnt[0] = 0
v, info = sla.minres(sla.LinearOperator((n + z, n + z), matvec=Kop), np.concatenate([rhs, np.zeros(z)]), rtol=tol2, maxiter=5000,
                     M=sla.LinearOperator((n + z, n + z), matvec=lambda r: Pinv * r))
print("MINRES augmented:  matvecs %d  relres of reduced system %.1e" % (cnt[0], np.linalg.norm(rhs - G(v[:n])) / np.linalg.norm(rhs)))
