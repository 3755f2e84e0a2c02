"""numpy restatement of the MINRES variant planned for the device (residual system from a warm start, block-diagonal preconditioner, true-residual
recursion, stopping on the REDUCED residual rho_x + w_z Az' rho_z) against Jacobi-PCG on the reduced system; 1/10 of config 3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
from scipy import sparse
import problem_gen as pg
from oracle import scs_oracle
sc = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-7
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
z = K["z"]; Rz = 1.0 / (1000.0 * scale); Rl = 1.0 / scale
ry = np.full(m, Rl); ry[:z] = Rz
Ar = A.tocsr(); Az = Ar[:z]; AzT = Az.T.tocsr()
vx = rng.standard_normal(n); vy = rng.standard_normal(m); ws = 0.3 * rng.standard_normal(n)
G = lambda x: rho_x * x + A.T @ ((A @ x) / ry)
b_red = rho_x * vx - A.T @ vy                      # reduced rhs
# ---- reference: Jacobi-PCG warm-started at ws
dG = rho_x + (A.multiply(A)).T @ (1.0 / ry)
def pcg():
    x = ws.copy(); r = b_red - G(x); zz = r / dG; p = zz.copy(); rz = r @ zz; it = 0
    lv = [1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6]; r0n = np.abs(r).max(); out = []
    while np.abs(r).max() >= tol and it < 20000:
        Gp = G(p); a = rz / (p @ Gp); x += a * p; r -= a * Gp; zz = r / dG; rz2 = r @ zz; p = zz + (rz2 / rz) * p; rz = rz2; it += 1
        while lv and np.abs(r).max() < lv[0] * r0n: out.append((lv.pop(0), it))
    print("PCG: steps to reach a fraction of the initial |r|_inf:", out)
    return x, it
x_cg, it_cg = pcg()
print("PCG: %d steps, |r_red|_inf %.2e" % (it_cg, np.abs(b_red - G(x_cg)).max()))
# ---- MINRES on the residual system  K d = [rho_x; 0]
y0 = vy + (A @ ws) / ry
rho_x0 = rho_x * (vx - ws) - A.T @ y0               # = EpiR0's r0
N = n + z
def Kop(v):
    x, yz = v[:n], v[n:]
    t = A @ x
    u = t / ry; u[:z] = yz
    return np.concatenate([A.T @ u + rho_x * x, t[:z] - Rz * yz])
dx = rho_x + (Ar[z:].multiply(Ar[z:])).T @ np.full(m - z, 1.0 / Rl)
dz = Rz + (Az.multiply(Az)) @ (1.0 / dx)
Minv = np.concatenate([1.0 / dx, 1.0 / dz])
b = np.concatenate([rho_x0, np.zeros(z)])
r1 = b.copy(); yp = Minv * r1; beta1 = np.sqrt(r1 @ yp)
oldb = 0.0; beta = beta1; dbar = 0.0; epsln = 0.0; phibar = beta1; cs = -1.0; sn = 0.0
w = np.zeros(N); w2 = np.zeros(N); d = np.zeros(N); r2 = r1.copy(); rho = b.copy()
it = 0
def red(rho): return rho[:n] + (AzT @ rho[n:]) / Rz
print("MINRES start: |r_red|_inf %.3e (PCG start %.3e)" % (np.abs(red(rho)).max(), np.abs(b_red - G(ws)).max()))
hist = []
lvm = [1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6]; outm = []; rn0 = np.abs(red(rho)).max()
while it < 5000:
    it += 1
    v = yp / beta
    y = Kop(v)
    if it >= 2: y -= (beta / oldb) * r1
    alfa = v @ y
    y -= (alfa / beta) * r2
    r1 = r2; r2 = y
    yp = Minv * r2
    oldb = beta; beta = np.sqrt(r2 @ yp)
    oldeps = epsln; delta = cs * dbar + sn * alfa; gbar = sn * dbar - cs * alfa; epsln = sn * beta; dbar = -cs * beta
    gamma = max(np.hypot(gbar, beta), 1e-300); cs = gbar / gamma; sn = beta / gamma; phi = cs * phibar; phibar = sn * phibar
    w1 = w2; w2 = w; w = (v - oldeps * w1 - delta * w2) / gamma
    d += phi * w
    rho = sn * sn * rho - phibar * cs * (r2 / beta)       # true residual b - K d, by recursion
    rn = np.abs(red(rho)).max()
    while lvm and rn < lvm[0] * rn0: outm.append((lvm.pop(0), it))
    if it % 40 == 0 or rn < tol:
        true = b - Kop(d)
        hist.append((it, rn, np.abs(red(true)).max(), np.abs(true - rho).max(), phibar))
    if rn < tol: break
print("MINRES: steps to reach a fraction of the initial |r_red|_inf:", outm)
for h in hist[-1:]: print("  it %4d  |r_red| recursion %.3e  true %.3e  |rho - true|_inf %.1e  phibar %.2e" % h)
x_m = ws + d[:n]
print("MINRES: %d steps; reduced residual of x: %.2e; |x - x_cg|_inf / |x_cg|_inf = %.2e" % (it, np.abs(b_red - G(x_m)).max(), np.abs(x_m - x_cg).max() / np.abs(x_cg).max()))
