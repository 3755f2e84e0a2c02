// dense.hpp — dense direct KKT solve on the device for SMALL problems (a direct linsys on the GPU;
// the reference's counterparts are its direct backends — QDLDL / cuDSS / the LAPACK "dense" module, R:meson.build:241-262,374-391,
// R:scs/py/__init__.py:28-37 — all absent from the snapshot or CPU / CUDA code).
//
// Why dense, and why an explicit inverse.  A lone config-5 problem (n = 1350, m = 4050, nnz = 54 000) spends its ADMM iteration in
// ~37 DEPENDENT launches of 2-5 us (3 + 4 per CG step), 0.2 ms per iteration whatever the GPU does in parallel — and that is the floor
// of a batch on any number of GPUs once only stragglers are left.  The reduced KKT matrix
//     G = R_x + P + A' R_y^{-1} A        (n x n, SPD; R_y takes two values that follow `scale`)
// of such a problem is 15 MB: small enough to keep G^{-1} itself in HBM (L2-resident for a lone problem: every XCD reads the same
// eighth of the columns in every iteration), so that the linear solve of an iteration becomes THREE dependent launches:
//     rhs = R_x v_x - A' v_y          (CSR-stream SpMV, epilogue EpiDenseRhs)
//     x   = G^{-1} rhs                (k_dense_gemv: one wavefront per column of the symmetric inverse, fixed shuffle tree)
//     y   = v_y + R_y^{-1} A x        (CSR-stream SpMV, epilogue EpiY)
// with no convergence flag to wait for: plain iterations are enqueued back to back without any host synchronisation.
// A triangular solve with a Cholesky factor would read half the bytes but is a chain of n / 64 dependent block steps — the latency
// this path exists to remove; G is SPD with a moderate condition number after equilibration (rho_x I + scale A'WA, m > n).
//
// G^{-1} is formed in place by a blocked Gauss-Jordan sweep without pivoting (SPD: every pivot block is a Schur complement, SPD
// again): per block step k one 64 x 64 pivot inverse in LDS (k_gj_pivot), the row panel P_k A_kJ and a copy of the old column panel
// (k_gj_panels), and a rank-64 update of ALL other tiles on v_mfma_f64_16x16x4_f64 (k_gj_update: C_IJ -= L_I R_J, the column panel
// becomes -L_I P_k).  3 launches x n / 64 steps, 2 n^3 flops, re-run at every adaptive-scale update (R_y changes).  All kernels are
// `d_X` bodies with a one-problem entry point and a batched one (blockIdx.y = member of a group, batch.hpp): one code, one
// arithmetic => a member of a grouped solve gets the bits of a solve of its own.  Deterministic: fixed accumulation orders, no atomics.
#pragma once
#include "common.hpp"
#include "psd.hpp"   // f64x4
#include "spmv.hpp"  // RDiag

namespace scship {

constexpr int kDenseB = 64;        // block size of the Gauss-Jordan sweep = tile edge of the update
constexpr int kDenseMaxN = 8192;   // G^{-1} of order 8192 is 537 MB (round 5; 4096 until round 4); the build kernel's two LDS columns are 128 KiB there
constexpr int kDenseThreads = 256;
constexpr int kDenseBuildThreads = 128;  // two wavefronts = two columns of G per workgroup, NP doubles of LDS each (<= 128 KiB of the CU's 160)
inline int dense_np(int n) { return (n + kDenseB - 1) / kDenseB * kDenseB; }

struct DenseMat {
  double *G;   // NP x NP, column-major, ld = NP: G, then (after the sweep) G^{-1}; rows / columns >= n are those of the identity
  double *Pk;  // 64 x 64: inverse of the current pivot block (symmetrised)
  double *L;   // NP x 64 (ld = NP): the column panel as it was before the step
  double *Rt;  // NP x 64 (ld = NP): the new row panel, transposed: Rt[j + NP kk] = (P_k A_kJ)[kk][j]
  int n, NP;
};
struct DenseSrc {  // what G is built from: CSR(A') = the caller's CSC(A), CSR(A), full symmetric CSR(P) or nullptr, diag_r (length n + m + 1)
  const int *at_rp, *at_ci;
  const double *at_v;
  const int *ar_rp, *ar_ci;
  const double *ar_v;
  const int *pf_rp, *pf_ci;
  const double *pf_v;
  const double *diag_r;
};

// rhs of the reduced system inside the ADMM iteration: rhs_j = R_x v_x,j - (A' v_y)_j      (cf. EpiR0 with ws = 0)
struct EpiDenseRhs {
  double *out;
  RDiag rx;
  const double *vx;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int j, double s, double *, double *) const { out[j] = rx[j] * vx[j] - s; }
};

// ---- G = R_x + P + A' R_y^{-1} A, one wavefront per column j: sum over the nonzeros (i, j) of column j, ascending i, of
// (a_ij a_ik) / r_i for every k of row i — products first, so that G_kj and G_jk get the same bits.
__device__ __forceinline__ void d_dense_build(DenseSrc S, DenseMat D) {
  extern __shared__ double dense_acc[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * (kDenseBuildThreads / 64) + wave;
  if (j >= D.NP) return;
  double *acc = dense_acc + (size_t)wave * D.NP;
  for (int i = lane; i < D.NP; i += 64) acc[i] = 0.;
  if (j < D.n) {
    for (int p = S.at_rp[j]; p < S.at_rp[j + 1]; ++p) {
      const int i = S.at_ci[p];
      const double aij = S.at_v[p], ri = S.diag_r[D.n + i];
      // (the lanes of one instruction hold distinct columns of row i; LDS operations of a wavefront execute in order, so the
      //  next row sees these sums)
      for (int q = S.ar_rp[i] + lane; q < S.ar_rp[i + 1]; q += 64) acc[S.ar_ci[q]] += (aij * S.ar_v[q]) / ri;
    }
    if (S.pf_rp)
      for (int q = S.pf_rp[j] + lane; q < S.pf_rp[j + 1]; q += 64) acc[S.pf_ci[q]] += S.pf_v[q];
    if (lane == 0) acc[j] += S.diag_r[j];
  } else if (lane == 0) {
    acc[j] = 1.;
  }
  for (int i = lane; i < D.NP; i += 64) D.G[(size_t)D.NP * j + i] = acc[i];
}

// ---- step k, part 1: P_k = inverse of the pivot block G[K, K] (64 x 64), unblocked in-place Gauss-Jordan.
// The block lives in REGISTERS: lane (r = tid & 63, w = tid >> 6) owns row r of the columns 16 w .. 16 w + 15 through all 64 steps; only
// the pivot row and column of a step go through LDS (published from the registers of their owners, double-buffered: one barrier per
// step), the loop is fully unrolled so that every register index is static.  (Round 4: the first version kept the block in LDS and
// made 16 dependent read-modify-write round trips per lane and step — 174 us per pivot block, 3.8 ms of a lone problem's 4.6 ms
// inversion; same operations on the same operands here.)
__device__ __forceinline__ void d_gj_pivot(DenseMat D, int k) {
  static_assert(kDenseB == 64 && kDenseThreads == 256, "ownership below: 64 rows x 4 column groups of 16");
  __shared__ double S[kDenseB][kDenseB + 1];
  __shared__ double colp[2][kDenseB], rowp[2][kDenseB];
  const int tid = threadIdx.x, K0 = k * kDenseB, r = tid & 63, w = tid >> 6;
  const size_t ld = (size_t)D.NP;
  double a[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) a[t] = D.G[(K0 + r) + ld * (K0 + 16 * w + t)];
#pragma unroll
  for (int p = 0; p < kDenseB; ++p) {
    const int b = p & 1, pw = p >> 4, pt = p & 15;
    if (w == pw) colp[b][r] = a[pt];
    if (r == p) {
#pragma unroll
      for (int t = 0; t < 16; ++t) rowp[b][16 * w + t] = a[t];
    }
    __syncthreads();
    const double d = 1.0 / rowp[b][p];
    const double cr = colp[b][r];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const double rc = rowp[b][16 * w + t];
      const bool cp = (w == pw) && (t == pt);  // this entry is in the pivot column
      double v;
      if (r == p) v = cp ? d : rc * d;
      else if (cp) v = -cr * d;
      else v = a[t] - cr * (rc * d);
      a[t] = v;
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) S[r][16 * w + t] = a[t];
  __syncthreads();
  for (int e = tid; e < kDenseB * kDenseB; e += kDenseThreads) {
    const int rr = e & 63, c = e >> 6;
    D.Pk[rr + kDenseB * c] = 0.5 * (S[rr][c] + S[c][rr]);
  }
}

// ---- step k, part 2 (workgroup J of NP / 64): L[J rows] = old G[J rows, K];  J != k: G[K, J] = P_k G[K, J], Rt[J rows] = its transpose
__device__ __forceinline__ void d_gj_panels(DenseMat D, int k) {
  __shared__ double Ps[kDenseB][kDenseB + 1], Bs[kDenseB][kDenseB + 1];
  const int tid = threadIdx.x, J = blockIdx.x, K0 = k * kDenseB, J0 = J * kDenseB;
  if (J0 >= D.NP) return;
  const size_t ld = (size_t)D.NP;
  for (int e = tid; e < kDenseB * kDenseB; e += kDenseThreads) {
    const int r = e & 63, c = e >> 6;
    D.L[(J0 + r) + ld * c] = D.G[(J0 + r) + ld * (K0 + c)];
  }
  if (J == k) return;
  for (int e = tid; e < kDenseB * kDenseB; e += kDenseThreads) {
    const int r = e & 63, c = e >> 6;
    Ps[r][c] = D.Pk[r + kDenseB * c];
    Bs[r][c] = D.G[(K0 + r) + ld * (J0 + c)];
  }
  __syncthreads();
  const int r = tid & 63, c0 = tid >> 6;  // outputs (r, c0 + 4 t): a wavefront shares c (LDS broadcast of Bs), r is conflict-free
  double out[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) out[t] = 0.;
  for (int kk = 0; kk < kDenseB; ++kk) {
    const double pr = Ps[r][kk];
#pragma unroll
    for (int t = 0; t < 16; ++t) out[t] += pr * Bs[kk][c0 + 4 * t];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 16; ++t) Ps[r][c0 + 4 * t] = out[t];  // (Ps is free now: the result, [row kk of the panel][column j])
  __syncthreads();
  for (int e = tid; e < kDenseB * kDenseB; e += kDenseThreads) {
    const int a = e & 63, b = e >> 6;
    D.G[(K0 + a) + ld * (J0 + b)] = Ps[a][b];  // a = panel row (fast index: down a column of G)
    D.Rt[(J0 + a) + ld * b] = Ps[b][a];        // a = column j of the panel (fast index: down a column of Rt)
  }
}

// ---- step k, part 3: tile (I, J) of NP/64 x NP/64 (blockIdx.x = I + nb J).  I != k, J != k: G_IJ -= L_I R_J;  J == k: G_Ik = -L_I P_k;
// (k, k): P_k;  row k otherwise: done by the panels kernel.  One wavefront = 16 rows i x 64 columns j: T[j][i] = sum_kk B[j][kk] A[i][kk]
// on v_mfma_f64_16x16x4_f64 with the operands swapped, so that a lane's results run DOWN a column of G (full 128-byte lines).
__device__ __forceinline__ void d_gj_update(DenseMat D, int k) {
  const int nb = D.NP / kDenseB;
  const int I = (int)blockIdx.x % nb, J = (int)blockIdx.x / nb;
  if (J >= nb) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const size_t ld = (size_t)D.NP;
  const int I0 = I * kDenseB, J0 = J * kDenseB;
  if (I == k) {
    if (J == k)
      for (int e = tid; e < kDenseB * kDenseB; e += kDenseThreads) D.G[(I0 + (e & 63)) + ld * (J0 + (e >> 6))] = D.Pk[e];
    return;
  }
  const bool colpanel = J == k;
  const double *Bop = colpanel ? D.Pk : D.Rt + J0;  // B[j][kk] at Bop[j + ldb kk]   (P_k is symmetric)
  const size_t ldb = colpanel ? (size_t)kDenseB : ld;
  // Round 4: the two 64 x 64 operand panels go through LDS, half of the k range at a time (two 16-byte loads per lane and panel,
  // every line read once per workgroup instead of once per wavefront and MFMA operand: the first version fetched 5 operands of
  // 8 bytes per lane for every 4 MFMAs straight from L2 and ran at 0.21 of the fp64 matrix peak).  Row stride 72: the four k-slices
  // a wavefront reads fall on different bank halves.  Same MFMAs in the same order: same bits.
  constexpr int KH = kDenseB / 2, LDT = kDenseB + 8;
  __shared__ double As[KH][LDT], Bs[KH][LDT];
  const int r2 = (tid & 31) * 2, c8 = tid >> 5;  // this lane stages rows r2, r2 + 1 of the columns c8 + 8 t
  const double *ga = D.L + (I0 + r2), *gb = Bop + r2;
  typedef double dbl2 __attribute__((ext_vector_type(2)));  // (HIP's double2 in an array lands in scratch memory)
  dbl2 va[4], vb[4];
  auto fetch_half = [&](int half) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = KH * half + c8 + 8 * t;
      va[t] = *reinterpret_cast<const dbl2 *>(ga + ld * c);
      vb[t] = *reinterpret_cast<const dbl2 *>(gb + ldb * c);
    }
  };
  auto stage_half = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      *reinterpret_cast<dbl2 *>(&As[c8 + 8 * t][r2]) = va[t];
      *reinterpret_cast<dbl2 *>(&Bs[c8 + 8 * t][r2]) = vb[t];
    }
  };
  f64x4 acc[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) acc[jt] = f64x4{0., 0., 0., 0.};
  fetch_half(0);
  // the tile's own entries are requested now and used after the products (unconditional: the column panel's are simply not used)
  double *gdst = D.G + (I0 + wave * 16 + li) + ld * (J0 + lk);
  double gold[4][4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int t = 0; t < 4; ++t) gold[jt][t] = gdst[ld * (jt * 16 + 4 * t)];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();  // every wavefront is done with the first half of the panels
    stage_half();
    if (half == 0) fetch_half(1);  // in flight while the first half is multiplied
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < KH; k0 += 4) {
      const double a = As[k0 + lk][wave * 16 + li];
      double b[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) b[jt] = Bs[k0 + lk][jt * 16 + li];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[jt], a, acc[jt], 0, 0, 0);
    }
  }
  // lane holds T[j = jt 16 + lk + 4 t][i = wave 16 + li]
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int t = 0; t < 4; ++t) gdst[ld * (jt * 16 + 4 * t)] = colpanel ? -acc[jt][t] : gold[jt][t] - acc[jt][t];
}

// ---- x = G^{-1} b: one wavefront per FOUR columns j of the (symmetric) inverse, y_j = sum_i Ginv[i][j] b_i, fixed order: a lane adds
// its rows 2 l, 2 l + 1, 2 l + 128, ... in ascending order, then the fixed shuffle tree.  (Round 4: 16-byte loads of the four
// columns against one pair of b entries — 6 memory instructions per 4 x 128 entries where one column per wavefront needed 16; a
// batch of 512 members streams 8 GB of inverses per lock-step iteration through this kernel.)
constexpr int kDenseGemvCols = 4;
inline int dense_gemv_blocks(int n) { return ceil_div(n, kDenseGemvCols * (kDenseThreads / 64)); }
__device__ __forceinline__ void d_dense_gemv(const double *__restrict__ Ginv, int NP, int n, const double *__restrict__ b, double *x,
                                             const int *stall) {
  SCS_STALL_GUARD(stall);
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j0 = (blockIdx.x * (kDenseThreads / 64) + wave) * kDenseGemvCols;
  if (j0 >= n) return;
  // (j0 + 3 < NP: NP is a multiple of 64 >= n; rows up to NP - 1 exist too: the padding is the identity)
  const double *c0 = Ginv + (size_t)NP * j0 + 2 * lane;
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  for (int i = 2 * lane; i < n; i += 128, c0 += 128) {
    const double b0 = b[i], b1 = i + 1 < n ? b[i + 1] : 0.;
    const dbl2 a0 = *reinterpret_cast<const dbl2 *>(c0), a1 = *reinterpret_cast<const dbl2 *>(c0 + NP);
    const dbl2 a2 = *reinterpret_cast<const dbl2 *>(c0 + 2 * (size_t)NP), a3 = *reinterpret_cast<const dbl2 *>(c0 + 3 * (size_t)NP);
    s0 += a0.x * b0; s0 += a0.y * b1;
    s1 += a1.x * b0; s1 += a1.y * b1;
    s2 += a2.x * b0; s2 += a2.y * b1;
    s3 += a3.x * b0; s3 += a3.y * b1;
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  s3 = wave_sum(s3);
  if (lane == 0) {
    x[j0] = s0;
    if (j0 + 1 < n) x[j0 + 1] = s1;
    if (j0 + 2 < n) x[j0 + 2] = s2;
    if (j0 + 3 < n) x[j0 + 3] = s3;
  }
}

// ---- x = G^{-1} b reading only the tiles on and below the diagonal (the inverse is symmetric; a batch of 512 config-5 members streams
// 8 GB of inverses per lock-step iteration through the full product — the HBM-bound part of that batch).  LAB SWITCH
// (SCS_HIP_DENSE_GEMV=half), not the default: a Gauss-Jordan inverse X is accurate on one side only — d_dense_gemv's X' b has the
// residual of || X G - I || ~ kappa eps, a product that mirrors one triangle of X sees its asymmetry, ~ kappa^2 eps (measured 1e-11
// vs 3.5e-8 at kappa = 2e4; ScsHipWork::dense_full_gemv).  Two launches:
//   d_dense_symv_tiles: workgroup t = tile (I, J), I >= J, of 64 x 64:  part[I][J] = T x_J,  part[J][I] = T' x_I  (I != J)
//   d_dense_symv_sum:   x_K = sum over o = 0 .. nb-1, ascending, of part[K][o]
// Every slot part[K][o] is written by exactly one tile; fixed summation orders everywhere => deterministic, and the same bits for a
// member of a group and a solve of its own.  (What is used of G^{-1} is its lower triangle: the operator is exactly symmetric.)
__device__ __forceinline__ void d_dense_symv_tiles(const double *__restrict__ Ginv, int NP, int n, const double *__restrict__ b, double *part,
                                                   const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double xs[2][kDenseB], red[4][kDenseB];
  const int nb = NP / kDenseB, tid = threadIdx.x;
  const int t = blockIdx.x;
  if (t >= nb * (nb + 1) / 2) return;
  int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  const int I0 = I * kDenseB, J0 = J * kDenseB;
  if (I0 >= n) return;  // (I >= J: padding rows and, with them or alone, padding columns: zero off the diagonal; their slots are never read — d_dense_symv_sum stops at o with o 64 < n)
  if (tid < 64) xs[0][tid] = J0 + tid < n ? b[J0 + tid] : 0.;
  else if (tid < 128) xs[1][tid - 64] = I0 + (tid - 64) < n ? b[I0 + (tid - 64)] : 0.;
  __syncthreads();
  const int r = tid & 63, q = tid >> 6;
  const double *src = Ginv + (size_t)(I0 + r) + (size_t)NP * (J0 + q * 16);
  double v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = src[(size_t)NP * k];
  double d = 0.;
#pragma unroll
  for (int k = 0; k < 16; ++k) d += v[k] * xs[0][q * 16 + k];
  red[q][r] = d;
  if (I != J) {
    const double xi = xs[1][r];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double s = wave_sum(v[k] * xi);  // (wavefront q holds all 64 rows of its 16 columns)
      if (r == 0) part[((size_t)J * nb + I) * kDenseB + q * 16 + k] = s;
    }
  }
  __syncthreads();
  if (tid < 64) part[((size_t)I * nb + J) * kDenseB + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}
__device__ __forceinline__ void d_dense_symv_sum(const double *__restrict__ part, int NP, int n, double *x, const int *stall) {
  SCS_STALL_GUARD(stall);
  const int nb = NP / kDenseB, no = (n + kDenseB - 1) / kDenseB;  // blocks o >= no hold only padding
  const int i = blockIdx.x * kDenseThreads + threadIdx.x;
  if (i >= n) return;
  const int K = i / kDenseB, r = i % kDenseB;
  const double *p = part + (size_t)K * nb * kDenseB + r;
  double s = 0.;
  for (int o = 0; o < no; ++o) s += p[(size_t)o * kDenseB];
  x[i] = s;
}
inline int dense_symv_tiles(int NP) { const int nb = NP / kDenseB; return nb * (nb + 1) / 2; }
inline size_t dense_symv_part_len(int NP) { const size_t nb = (size_t)(NP / kDenseB); return nb * nb * kDenseB; }

__global__ __launch_bounds__(kDenseThreads) void k_dense_symv_tiles(const double *__restrict__ Ginv, int NP, int n, const double *__restrict__ b,
                                                                    double *part, const int *stall) {
  d_dense_symv_tiles(Ginv, NP, n, b, part, stall);
}
__global__ __launch_bounds__(kDenseThreads) void k_dense_symv_sum(const double *__restrict__ part, int NP, int n, double *x, const int *stall) {
  d_dense_symv_sum(part, NP, n, x, stall);
}
// x = G^{-1} b on stream s (two launches)
inline void dense_apply(const double *Ginv, int NP, int n, const double *b, double *part, double *x, const int *stall, hipStream_t s) {
  hipLaunchKernelGGL(k_dense_symv_tiles, dim3(dense_symv_tiles(NP)), dim3(kDenseThreads), 0, s, Ginv, NP, n, b, part, stall);
  hipLaunchKernelGGL(k_dense_symv_sum, dim3(ceil_div(n, kDenseThreads)), dim3(kDenseThreads), 0, s, (const double *)part, NP, n, x, stall);
}

__global__ __launch_bounds__(kDenseBuildThreads) void k_dense_build(DenseSrc S, DenseMat D) { d_dense_build(S, D); }
__global__ __launch_bounds__(kDenseThreads) void k_gj_pivot(DenseMat D, int k) { d_gj_pivot(D, k); }
__global__ __launch_bounds__(kDenseThreads) void k_gj_panels(DenseMat D, int k) { d_gj_panels(D, k); }
__global__ __launch_bounds__(kDenseThreads) void k_gj_update(DenseMat D, int k) { d_gj_update(D, k); }
__global__ __launch_bounds__(kDenseThreads) void k_dense_gemv(const double *__restrict__ Ginv, int NP, int n, const double *__restrict__ b,
                                                              double *x, const int *stall) {
  d_dense_gemv(Ginv, NP, n, b, x, stall);
}
// batched entry points of the factorisation (the step index k is a launch argument, so these are not k_grouped instances):
// blockIdx.y = position in `list`, tab[list[.]] = the member's matrices
__global__ __launch_bounds__(kDenseBuildThreads) void k_dense_build_g(const DenseSrc *src, const DenseMat *tab, const int *list) {
  d_dense_build(src[list[blockIdx.y]], tab[list[blockIdx.y]]);
}
__global__ __launch_bounds__(kDenseThreads) void k_gj_pivot_g(const DenseMat *tab, const int *list, int k) { d_gj_pivot(tab[list[blockIdx.y]], k); }
__global__ __launch_bounds__(kDenseThreads) void k_gj_panels_g(const DenseMat *tab, const int *list, int k) { d_gj_panels(tab[list[blockIdx.y]], k); }
__global__ __launch_bounds__(kDenseThreads) void k_gj_update_g(const DenseMat *tab, const int *list, int k) { d_gj_update(tab[list[blockIdx.y]], k); }

inline size_t dense_build_lds(int NP) { return sizeof(double) * (size_t)NP * (kDenseBuildThreads / 64); }

// the whole factorisation of ONE problem on stream s
inline void dense_factor(const DenseSrc &S, const DenseMat &D, hipStream_t s) {
  const int nb = D.NP / kDenseB;
  hipLaunchKernelGGL(k_dense_build, dim3(ceil_div(D.NP, kDenseBuildThreads / 64)), dim3(kDenseBuildThreads), dense_build_lds(D.NP), s, S, D);
  for (int k = 0; k < nb; ++k) {
    hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(kDenseThreads), 0, s, D, k);
    hipLaunchKernelGGL(k_gj_panels, dim3(nb), dim3(kDenseThreads), 0, s, D, k);
    hipLaunchKernelGGL(k_gj_update, dim3(nb * nb), dim3(kDenseThreads), 0, s, D, k);
  }
}
// ... of the listed members of a group (tables and list in device memory)
inline void dense_factor_group(const DenseSrc *src, const DenseMat *tab, const int *list, int count, int NP, hipStream_t s) {
  if (count <= 0) return;
  const int nb = NP / kDenseB;
  hipLaunchKernelGGL(k_dense_build_g, dim3(ceil_div(NP, kDenseBuildThreads / 64), count), dim3(kDenseBuildThreads), dense_build_lds(NP), s,
                     src, tab, list);
  for (int k = 0; k < nb; ++k) {
    hipLaunchKernelGGL(k_gj_pivot_g, dim3(1, count), dim3(kDenseThreads), 0, s, tab, list, k);
    hipLaunchKernelGGL(k_gj_panels_g, dim3(nb, count), dim3(kDenseThreads), 0, s, tab, list, k);
    hipLaunchKernelGGL(k_gj_update_g, dim3(nb * nb, count), dim3(kDenseThreads), 0, s, tab, list, k);
  }
}

}  // namespace scship
