// minres.hpp — the indirect KKT solve of an ADMM iteration by preconditioned MINRES on the system with the ZERO-CONE block left
// un-eliminated (north_star: "CG/MINRES over A and A' in CSC"; the role of scs_source/linsys/cpu/indirect/private.c, R:meson.build:261,
// and of its GPU twin R:meson.build:303-304 — absent).
//
// Why: the reduced system PCG works on, G = R_x + P + A' R_y^{-1} A, weighs the z rows of a zero cone 1000 x heavier than the others
// (R_y = 1 / (1000 scale) there, as in the reference).  With 10 % zero-cone rows (BASELINE config 3) the Jacobi-preconditioned CG needs
// ~330 steps per ADMM iteration: the z heavy directions are a rank-z bump of the spectrum.  Leaving those rows in the system,
//     K [x; y_z] = [[R_x + P + A_l' R_l^{-1} A_l,  A_z'], [A_z,  -R_z]] [x; y_z] = [rhs_x; rhs_z],
// keeps every block O(1): symmetric, indefinite — MINRES — and a block-diagonal preconditioner (diag of the (1,1) block; diag of the
// Schur complement R_z + A_z D_x^{-1} A_z') is enough.  One application of K costs the same two products over ALL of A as one of G
// (u = R_l^{-1} A x on the other rows, u_z = y_z on the zero-cone rows; A' u).  tools/dbg/minres_algo_proto.py (round 4, numpy, 1/10 of
// config 3): 1.7 x fewer steps than PCG at the accuracy the ADMM loop asks for.
//
// What is solved is the RESIDUAL system from the warm start ws (K d = [r0; 0], r0 = the reduced residual the fused CG start already
// forms: EpiR0), x = ws + d_x at the end, then y = v_y + R_y^{-1} A x as everywhere else.  The stopping test is the reference's: the
// inf-norm of the residual OF THE REDUCED SYSTEM, rho_x + A_z' rho_z / R_z (rho = b - K d by its own recursion), against the same
// tolerance S_TOL — MINRES's own preconditioned residual norm is 1e6 x optimistic here (the y_z block hides the 1 / R_z amplification).
//
// A step is six launches, all scalars on the device (two banks, written by block 0 for the NEXT step, read by everybody now):
//   K1' (A,  EpiMrU)  u = R_l^{-1} A v_x | v_z,  y_z = (A v_x)_z - R_z v_z - c1 r1_z,  partial v_z . y_z            v = M^{-1} r2 / beta
//   K2' (A', EpiMrY)  y_x = A' u + R_x v_x (+ P v_x) - c1 r1_x,  partial v_x . y_x
//   k_mr_v1           alfa;  r3 = y - (alfa / beta) r2;  yp' = M^{-1} r3;  partial r3 . yp'
//   k_mr_v2           beta', the Givens rotation;  w' = (v - epsln w1 - delta w2) / gamma;  d += phi w';  rho = sn^2 rho - phibar cs r3 / beta'
//   k_mr_red          partial max |rho_x + A_z' rho_z / R_z|     (A_z': the zero-cone rows' prefix of every column, own small CSR)
//   k_mr_fin          F_ITERS, S_RNORM, F_DONE
// Deterministic: fixed-order reductions, no atomics.  Vectors have length N = n + z, x part first.
#pragma once
#include "vec.hpp"
#include "spmv.hpp"

namespace scship {

// scalar block: [0] alfa of the step in flight; banks (step parity) at kMrBank0 + 8 b:
//   [0] beta  [1] c1 = beta / oldb (0 at the first step)  [2] cs  [3] sn  [4] dbar  [5] epsln  [6] phibar
enum : int { MR_ALFA = 0, kMrBank0 = 8, kMrBankLen = 8, kMrScalars = 32 };
enum : int { MRB_BETA = 0, MRB_C1, MRB_CS, MRB_SN, MRB_DBAR, MRB_EPSLN, MRB_PHIBAR };

// ---- preconditioner: M^{-1} = [1 / diag(R_x + P + A_l' R_l^{-1} A_l);  1 / diag(R_z + A_z D_x^{-1} A_z')] ----
// At = CSR(A') (row j = column j of A: sorted row indices), Ar = CSR(A)
__global__ __launch_bounds__(kVecThreads) void k_mr_precond_x(const int *__restrict__ colptr, const int *__restrict__ rowidx,
                                                              const double *__restrict__ val, RDiag rx, RDiag ry, const double *__restrict__ Pdiag,
                                                              int n, int z, double *Minv) {
  const int j = blockIdx.x * kVecThreads + threadIdx.x;
  if (j >= n) return;
  double d = rx[j] + (Pdiag ? Pdiag[j] : 0.);
  for (int k = colptr[j]; k < colptr[j + 1]; ++k) {
    const int i = rowidx[k];
    if (i >= z) d += val[k] * val[k] / ry[i];
  }
  Minv[j] = 1. / d;
}
__global__ __launch_bounds__(kVecThreads) void k_mr_precond_z(const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                              const double *__restrict__ val, RDiag ry, int n, int z, double *Minv) {
  const int i = blockIdx.x * kVecThreads + threadIdx.x;
  if (i >= z) return;
  double d = ry[i];
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) d += val[k] * val[k] * Minv[colidx[k]];
  Minv[n + i] = 1. / d;
}

// ---- A_z' as its own CSR (n rows; the entries of column j of A with row index < z: a prefix, the indices are sorted) ----
__global__ __launch_bounds__(kVecThreads) void k_mr_azt_count(const int *__restrict__ colptr, const int *__restrict__ rowidx, int n, int z, int *cnt) {
  const int j = blockIdx.x * kVecThreads + threadIdx.x;
  if (j >= n) return;
  int lo = colptr[j], hi = colptr[j + 1];
  const int b = lo;
  while (lo < hi) {  // first entry with row index >= z
    const int mid = (lo + hi) >> 1;
    if (rowidx[mid] < z) lo = mid + 1;
    else hi = mid;
  }
  cnt[j] = lo - b;
}
__global__ __launch_bounds__(kVecThreads) void k_mr_azt_fill(const int *__restrict__ colptr, const int *__restrict__ rowidx, const double *__restrict__ val,
                                                             int n, const int *__restrict__ zptr, int *zidx, double *zval) {
  const int j = blockIdx.x * kVecThreads + threadIdx.x;
  if (j >= n) return;
  const int b = colptr[j], o = zptr[j], c = zptr[j + 1] - o;
  for (int k = 0; k < c; ++k) {
    zidx[o + k] = rowidx[b + k];
    zval[o + k] = val[b + k];
  }
}

// ---- the two products of a step ----
struct EpiMrU {  // on A (row i of m): u, the z block of K v, its part of v . (K v - c1 r1)
  double *u;              // m
  double *Y;              // N: the z block goes to Y[n + i]
  const double *yp;       // N: beta v
  const double *r1;       // N
  const double *bank;     // this step's scalars
  RDiag ry;
  int n, z;
  double *partial;
  static constexpr int kSums = 1, kMaxs = 0;
  __device__ void operator()(int i, double s, double *sums, double *) const {
    const double ib = 1. / bank[MRB_BETA];
    const double t = s * ib;
    if (i < z) {
      const double vz = yp[n + i] * ib;
      const double yz = t - ry[i] * vz - bank[MRB_C1] * r1[n + i];
      u[i] = vz;
      Y[n + i] = yz;
      sums[0] += vz * yz;
    } else {
      u[i] = t / ry[i];
    }
  }
};
struct EpiMrY {  // on A' (row j of n): the x block of K v - c1 r1, its part of the dot product
  double *Y, *Y2;         // Y2: second half's raw partial sums on a split layout (nullable otherwise)
  const double *yp, *r1;
  const double *bank;
  RDiag rx;
  int has_P;              // Y[j] holds (P yp_x)[j] on entry
  double *partial;
  static constexpr int kSums = 1, kMaxs = 0;
  __device__ void operator()(int j, double s, double *sums, double *) const {
    const double ib = 1. / bank[MRB_BETA];
    const double vx = yp[j] * ib;
    double y = s + rx[j] * vx - bank[MRB_C1] * r1[j];
    if (has_P) y += Y[j] * ib;
    Y[j] = y;
    sums[0] += vx * y;
  }
  // linear in the row sum: the first half carries the other terms, the second its raw sum (k_mr_v1 reads Y + Y2)
  __device__ void split(int j, double s, int part, double *sums, double *maxs) const {
    if (part == 0) {
      (*this)(j, s, sums, maxs);
    } else {
      Y2[j] = s;
      sums[0] += yp[j] / bank[MRB_BETA] * s;
    }
  }
};

// ---- start: b = [r0; 0], r2 = b, yp = M^{-1} b, w = 0, d = 0, rho = b; partial b . yp ----
// (runs whether or not the solve is already done: the finish adds d)
__global__ __launch_bounds__(kVecThreads) void k_mr_init(const double *__restrict__ r0, const double *__restrict__ Minv, int n, long N, double *r1, double *r2,
                                                         double *yp, double *w1, double *w2, double *d, double *rho, double *part, const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double sm[kVecThreads / 64];
  double s = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < N; i += (long)gridDim.x * kVecThreads) {
    const double b = i < n ? r0[i] : 0.;
    const double y = Minv[i] * b;
    r1[i] = 0.;  // (multiplied by c1 = 0 in the first step: must be finite)
    r2[i] = b;
    yp[i] = y;
    w1[i] = 0.;
    w2[i] = 0.;
    d[i] = 0.;
    rho[i] = b;
    s += b * y;
  }
  s = block_sum<kVecThreads>(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(kVecThreads) void k_mr_fin0(const double *part, int np, double *mr, const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double sm[kVecThreads / 64];
  const double s = part_sum(part, np, sm);
  if (threadIdx.x == 0) {
    const double beta1 = sqrt(fmax(s, 0.));
    double *b = mr + kMrBank0;
    b[MRB_BETA] = beta1 > 0. ? beta1 : 1.;  // (a zero right-hand side never gets here: F_DONE)
    b[MRB_C1] = 0.;
    b[MRB_CS] = -1.;
    b[MRB_SN] = 0.;
    b[MRB_DBAR] = 0.;
    b[MRB_EPSLN] = 0.;
    b[MRB_PHIBAR] = beta1;
  }
}

// ---- r3 = y - (alfa / beta) r2, yp' = M^{-1} r3 ----
__global__ __launch_bounds__(kVecThreads) void k_mr_v1(const double *__restrict__ Y, const double *__restrict__ Y2, const double *__restrict__ r2,
                                                       const double *__restrict__ Minv, int n, long N, double *r3, double *ypn,
                                                       const double *partA, int nA, const double *partB, int nB, double *mr, int bank,
                                                       double *part, const int *fl) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc;
  {
    const double a = part_sum(partA, nA, sm);
    const double b = part_sum(partB, nB, sm);
    if (threadIdx.x == 0) {
      bc = a + b;
      if (blockIdx.x == 0) mr[MR_ALFA] = bc;
    }
    __syncthreads();
  }
  const double c2 = bc / mr[kMrBank0 + kMrBankLen * bank + MRB_BETA];
  double s = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < N; i += (long)gridDim.x * kVecThreads) {
    double y = Y[i];
    if (Y2 && i < n) y += Y2[i];
    y -= c2 * r2[i];
    const double p = Minv[i] * y;
    r3[i] = y;
    ypn[i] = p;
    s += y * p;
  }
  s = block_sum<kVecThreads>(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---- the rotation, the direction, the iterate and the true residual ----
__global__ __launch_bounds__(kVecThreads) void k_mr_v2(const double *__restrict__ yp, const double *__restrict__ w1, const double *__restrict__ w2,
                                                       double *wn, double *d, double *rho, const double *__restrict__ r3, long N,
                                                       const double *partV, int nV, double *mr, int bank, const int *fl) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc[7];
  {
    const double bn2 = part_sum(partV, nV, sm);
    if (threadIdx.x == 0) {
      const double *b = mr + kMrBank0 + kMrBankLen * bank;
      const double beta = b[MRB_BETA], cs = b[MRB_CS], sn = b[MRB_SN], dbar = b[MRB_DBAR], epsln = b[MRB_EPSLN], phibar = b[MRB_PHIBAR];
      const double alfa = mr[MR_ALFA];
      const double betan = sqrt(fmax(bn2, 0.));
      const double oldeps = epsln;
      const double delta = cs * dbar + sn * alfa;
      const double gbar = sn * dbar - cs * alfa;
      const double gamma = fmax(sqrt(gbar * gbar + betan * betan), 1e-300);
      const double csn = gbar / gamma, snn = betan / gamma;
      const double phi = csn * phibar, phibarn = snn * phibar;
      bc[0] = 1. / beta;
      bc[1] = oldeps;
      bc[2] = delta;
      bc[3] = 1. / gamma;
      bc[4] = phi;
      bc[5] = betan > 0. ? phibarn * csn / betan : 0.;
      bc[6] = snn * snn;
      if (blockIdx.x == 0) {
        double *o = mr + kMrBank0 + kMrBankLen * (bank ^ 1);
        o[MRB_BETA] = betan > 0. ? betan : 1.;
        o[MRB_C1] = betan / beta;
        o[MRB_CS] = csn;
        o[MRB_SN] = snn;
        o[MRB_DBAR] = -cs * betan;
        o[MRB_EPSLN] = sn * betan;
        o[MRB_PHIBAR] = phibarn;
      }
    }
    __syncthreads();
  }
  const double ib = bc[0], oldeps = bc[1], delta = bc[2], ig = bc[3], phi = bc[4], cr = bc[5], sn2 = bc[6];
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < N; i += (long)gridDim.x * kVecThreads) {
    const double w = (yp[i] * ib - oldeps * w1[i] - delta * w2[i]) * ig;
    wn[i] = w;
    d[i] += phi * w;
    rho[i] = sn2 * rho[i] - cr * r3[i];
  }
}

// ---- the reduced residual rho_x + A_z' rho_z / R_z: partial inf-norms (thread per row of A_z') ----
__global__ __launch_bounds__(kVecThreads) void k_mr_red(const int *__restrict__ zptr, const int *__restrict__ zidx, const double *__restrict__ zval,
                                                        const double *__restrict__ rho, int n, RDiag ry, double *part, const int *fl) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  double mx = 0.;
  for (int j = blockIdx.x * kVecThreads + threadIdx.x; j < n; j += gridDim.x * kVecThreads) {
    double s = 0.;
    for (int k = zptr[j]; k < zptr[j + 1]; ++k) s += zval[k] * rho[n + zidx[k]] / ry[zidx[k]];
    mx = fmax(mx, abs_nan_inf(rho[j] + s));
  }
  mx = block_max<kVecThreads>(mx, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = mx;
}
__global__ __launch_bounds__(kVecThreads) void k_mr_fin(const double *part, int np, double *sc, int *fl, double tolf) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  const double rn = part_max(part, np, sm);
  if (threadIdx.x == 0) {
    sc[S_RNORM] = rn;
    fl[F_ITERS] += 1;
    if (rn < tolf * sc[S_TOL]) fl[F_DONE] = 1;
  }
}

// ---- x = ws + d_x (the y block follows from x as in the CG start: EpiY) ----
__global__ __launch_bounds__(kVecThreads) void k_mr_x(double *x, const double *__restrict__ ws, const double *__restrict__ d, int n, const int *fl,
                                                      const int *stall) {
  SCS_STALL_GUARD(stall);
  if (fl[F_ZERO_RHS]) return;  // k_fin_head has set the zero solution
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) x[i] = ws[i] + d[i];
}

}  // namespace scship
