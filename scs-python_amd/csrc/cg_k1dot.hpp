// cg_k1dot.hpp — the PCG step of the large-matrix path with K2 reduced to a pure A' product (round 4).
//
// CG needs p'Gp with G = R_x + A' R_y^{-1} A.  Rounds 1-3 formed it in K2's epilogue (Gp_j = (A'z)_j + r_x,j p_j and the partial sums
// of p_j Gp_j), which makes K2 read p — 16 MB per launch on the split layout (both workgroups of a row chunk), +5 % HBM traffic and
// the reason K2 ran ~5 us behind K1 (profiles/r04_spmv_pmc.txt).  But
//     p'Gp = p' R_x p + (A p)' R_y^{-1} (A p) = sum_j r_x,j p_j^2 + sum_i s_i z_i,      s = A p,  z = R_y^{-1} s,
// and K1 has s_i and z_i in hand when it stores z_i: the second sum costs one register accumulator in K1's epilogue (EpiDivRDot), the
// first one a partial sum in the kernel that forms p (k_cg_dir_pp; k_pp_part behind the fused CG start).  K2 then stores raw row sums
// (EpiAtRaw: nothing read but the pass stream and the gathered z), and the R_x p term of Gp moves into the CG update, which reads p
// anyway (x += alpha p).  Same mathematics; a different (fixed) summation order of p'Gp than the path that keeps K2's dot, which stays
// in use for small problems (CSR-stream layouts, grouped solves, the persistent kernel) and for QPs — selected per workspace
// (ScsHipWork::k1dot), never mixed inside one linear solve.
// MEASURED (round 4, metric workload, three A/B pairs): K2 90.0 -> 86.4 us, K1 85.3 -> 86.5 us (its block reduction), i.e. both at
// 0.39-0.40 of the HBM peak — but ADMM iterations/s 310-312 -> 305-306 and the steady window 507-511 -> 491-496: the second reduction
// chain costs more than the 16 MB saved.  The product keeps the dot in K2; this path is an opt-in (SCS_HIP_K1DOT=1) kept under test
// (tests/test_hip_fullsize.py) as the record of the experiment.
#pragma once
#include "spmv.hpp"
#include "vec.hpp"

namespace scship {

struct EpiDivRDot {  // z[r] = s / ry[r];  partial sums of s z  (= the A' R_y^{-1} A part of p'Gp)
  double *z;
  RDiag ry;
  double *partial;
  static constexpr int kSums = 1, kMaxs = 0;
  __device__ void operator()(int r, double s, double *sums, double *) const {
    const double zr = s / ry[r];
    z[r] = zr;
    sums[0] += s * zr;
  }
};

struct EpiAtRaw {  // Gp[r] = s (first half / whole row), Gp2[r] = s (second half of a split layout): raw row sums of A'z, no reduction
  double *Gp, *Gp2;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { Gp[r] = s; }
  __device__ void split(int r, double s, int part, double *, double *) const { (part ? Gp2 : Gp)[r] = s; }
};

// partial sums of r_x,j p_j^2 (behind the CG start, which forms p0 = M r0 inside an SpMV epilogue)
__global__ __launch_bounds__(kVecThreads) void k_pp_part(const double *__restrict__ p, RDiag rx, int n, double *part, const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double sm[kVecThreads / 64];
  double s = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) s += rx[(int)i] * p[i] * p[i];
  s = block_sum<kVecThreads>(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// k_cg_update with p'Gp = sum(k1_part) + sum(pp_part) and the R_x p term of Gp applied here
__global__ __launch_bounds__(kVecThreads) void k_cg_update_k1dot(double *x, double *r, const double *__restrict__ p, const double *__restrict__ Gp,
                                                                 const double *__restrict__ Gp2, const double *__restrict__ M, int n, double *yacc,
                                                                 const double *__restrict__ z, int m, const double *k1_part, int k1_np,
                                                                 const double *pp_part, int pp_np, RDiag rx, double *sc, const int *fl, double *part) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc;
  {  // one fixed-order reduction over both partial arrays (only their total is needed)
    double t = 0.;
    for (int i = threadIdx.x; i < k1_np; i += kVecThreads) t += k1_part[i];
    for (int i = threadIdx.x; i < pp_np; i += kVecThreads) t += pp_part[i];
    t = block_sum<kVecThreads>(t, sm);
    if (threadIdx.x == 0) {
      bc = sc[(fl[F_STEP] & 1) ? S_ZTR_B : S_ZTR] / t;
      if (blockIdx.x == 0) sc[S_ALPHA] = bc;
    }
    __syncthreads();
  }
  const double alpha = bc;
  const int b = (int)blockIdx.x, nb = (int)gridDim.x, tid = (int)threadIdx.x;
  double mx = 0., s = 0.;
  if (yacc)
    for (long i = (long)b * kVecThreads + tid; i < m; i += (long)nb * kVecThreads) yacc[i] += alpha * z[i];
  for (long i = (long)b * kVecThreads + tid; i < n; i += (long)nb * kVecThreads) {
    const double pi = p[i];
    x[i] += alpha * pi;
    const double ri = r[i] - alpha * ((Gp2 ? Gp[i] + Gp2[i] : Gp[i]) + rx[(int)i] * pi);
    r[i] = ri;
    mx = fmax(mx, abs_nan_inf(ri));
    s += (M[i] * ri) * ri;
  }
  mx = block_max<kVecThreads>(mx, sm);
  s = block_sum<kVecThreads>(s, sm);
  if (tid == 0) {
    part[b] = mx;
    part[nb + b] = s;
  }
}

// k_cg_dir that also leaves the partial sums of r_x p^2 of the NEW direction (for the next step's alpha)
__global__ __launch_bounds__(kVecThreads) void k_cg_dir_pp(double *p, const double *__restrict__ r, const double *__restrict__ M, int n,
                                                           const double *upd_part, int upd_np, RDiag rx, double *pp_part, double *sc, int *fl) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc[2];
  const int slot = fl[F_STEP] & 1;
  {
    const double rn = part_max(upd_part, upd_np, sm);
    const double ztr = part_sum(upd_part + upd_np, upd_np, sm);
    if (threadIdx.x == 0) {
      bc[0] = ztr / sc[slot ? S_ZTR_B : S_ZTR];
      bc[1] = ztr;
      if (blockIdx.x == 0) {
        sc[S_RNORM] = rn;
        sc[S_BETA] = bc[0];
        sc[slot ? S_ZTR : S_ZTR_B] = ztr;
        fl[F_ITERS] += 1;
        if (rn < sc[S_TOL]) fl[F_DONE] = 1;
      }
    }
    __syncthreads();
  }
  const double beta = bc[0];
  double s = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) {
    const double pn = M[i] * r[i] + beta * p[i];
    p[i] = pn;
    s += rx[(int)i] * pn * pn;
  }
  s = block_sum<kVecThreads>(s, sm);
  if (threadIdx.x == 0) pp_part[blockIdx.x] = s;
}

}  // namespace scship
