// psd.hpp — K9: batched projection onto the PSD cone on the fp64 matrix cores.
//
// Plays the role of the LAPACK syev* path of scs_source/src/cones.c under
// USE_LAPACK (R:meson.build:145-147,188; absent).  Vector layout per cone:
// lower triangle, column-major, off-diagonals scaled by sqrt(2)
// (R:test/gen_random_cone_prob.py:153-173, R:test/test_scs_coverage.py:1387-1393).
// The PSD cone is self-dual, so Pi_{K*} = Pi_K.
//
// One 1024-lane workgroup (16 wavefronts) per matrix; all matrices of the cone
// run concurrently, one CU each.  Eigen-decomposition: two-sided BLOCK Jacobi,
// block size 8, blocks paired in the round-robin (tournament) parallel ordering.
// One outer step with NB/2 disjoint block pairs (p,q):
//   phase 1  each pair's 16x16 pivot [[App,Apq],[Aqp,Aqq]] is diagonalised by ONE wavefront
//            (scalar parallel-order Jacobi held in LDS, 8 rotations per inner step, each
//            (k,k') rotation pair owning one 2x2 block) -> 16x16 orthogonal W_k;
//   phase 2  A <- W' A W and V <- V W as 16x16x16 products on v_mfma_f64_16x16x4_f64:
//            every (k <= k') pair of block pairs owns the 16x16 block rows{p,q} x cols{p',q'},
//            computes W_k' (B W_k') with 8 MFMAs and writes it and its mirror (A stays exactly
//            symmetric, the update is in place); V row tiles likewise.
// The C/D register layout of the f64 MFMA (row = (lane>>4) + 4*reg, col = lane&15) is exactly
// its B-operand layout for k-step `reg`, so the intermediate product feeds the second MFMA
// chain straight from registers.  A and V live in an L2-resident scratch; finally
// X+ = (V sqrt(L+)) (V sqrt(L+))' over lower-triangular 16x16 tiles, again on MFMA.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "cones.hpp"

#ifndef PSD_INNER
#define PSD_INNER 1
#endif
#ifndef PSD_PROFILE
#define PSD_PROFILE 0  // 1 (tools/psd_lab.hip): thread 0 accumulates 100 MHz wall-clock ticks per phase into state[1..7]
#endif
#if PSD_PROFILE
#define PSD_TICK(var) const long long var = (threadIdx.x == 0) ? wall_clock64() : 0
#define PSD_ACC(slot, a, b) do { if (threadIdx.x == 0) prof[slot] += (double)((b) - (a)); } while (0)
#else
#define PSD_TICK(var) do { } while (0)
#define PSD_ACC(slot, a, b) do { } while (0)
#endif

namespace scship {

constexpr int kPsdThreads = 1024;
constexpr int kPsdWaves = kPsdThreads / 64;
constexpr int kPsdMaxSweeps = 30;
constexpr int kPsdB = 8;  // block size; pivots are 2*kPsdB = 16 = one MFMA tile
#ifndef PSD_LD
#define PSD_LD 17
#endif
constexpr int kPsdLd = PSD_LD;  // leading dimension of the 16x16 pivot S and rotation W in LDS: stride 16 would put a whole
                            // row of the 2x2-block accesses (and of the MFMA operand loads of W) on 2 banks
#ifndef PSD_WLD
#define PSD_WLD 17
#endif
constexpr int kPsdWLd = PSD_WLD;      // leading dimension of W
constexpr int kPsdWsz = 16 * 17;  // 272 doubles reserved per S / W
constexpr int kPsdWaveLds = 2 * kPsdWsz;  // per wave: S (also the 16x17 transpose scratch), W
constexpr int kPsdWarmPeriod = 32;  // calls between two re-orthogonalisations of the warm-start basis V
constexpr int kPsdMaxH = 512;  // pivots per step: order <= 8192 (LDS schedule arrays; an order-8192 matrix needs 2.1 GB of scratch)
#ifndef PSD_OFFTOL2
#define PSD_OFFTOL2 1e-16
#endif
constexpr double kPsdOffTol2 = PSD_OFFTOL2;  // sweeps stop at ||offdiag||_F^2 <= this * ||A||_F^2 (see the reconstruction)
// Inside the ADMM loop — while the Anderson history is still filling, i.e. in the first lookback x interval iterations (always, without
// acceleration) — the stopping level follows the residuals, as the tolerance of the inexact linear solve does (vec.hpp k_fin_head):
// tol2 points at the iteration's P_PSD_TOL2 = clamp(level, 1e-8, 1e-3)^2, level = min(1e-2 * primal/dual residual, certificate
// residuals) of the last convergence check (work.hpp note_check_residuals / psd_tol2_of).  While the iterate is far from the
// solution an eigen-decomposition to 1e-8 buys nothing: the second-order reconstruction leaves O(|E|^3) of the remaining off-diagonal
// part E (~5e-9 relative at |E| = 1e-4, measured on the special-spectra tests), orders below the residual it is tied to; from
// residual 1e-6 down the level IS the fixed 1e-8.  Measured (round 3): config 4, iterations 5..105: 3 sweeps per projection -> 1-2,
// cone pass 1.8 -> 1.18 ms, 374 -> 495 iterations/s; a whole config-4 solve to 1e-4: 750 iterations in 1.5 s; the 512 config-5
// problems (5 small PSD cones each): total iterations unchanged to 1 %.  Why it is this timid — factor 1e-2, certificate residuals
// at full weight, off once Anderson extrapolates:
//  * factor 1e-1 (551 iterations/s): the golden SDP `feas0` at eps 1e-9, an accelerated solve whose iteration count is chaotic anyway
//    (650 .. 1050 under 1e-9 perturbations of alpha), took 7000 iterations instead of 750 (23 700 when the level stayed on for the
//    whole solve); with 1e-2 its worst observed run is 1900 (tools/dbg/psd_tol_effect.py, psd_tol_chaos.py);
//  * a problem drifting to an infeasibility / unboundedness certificate has a LARGE primal/dual residual for good: tied to that alone
//    the sweeps stayed at 1e-3 and a golden infeasible instance never met eps_infeas (tools/dbg/psd_tol_infeas.py: all 18 golden
//    certificate cases x the acceleration variants terminate as with the fixed level now);
//  * plain ADMM tolerates inexact projections, the secant model of the acceleration does not (type-II steps at interval 1 stalled).
// SCS_HIP_PSD_TOL=fixed turns the coupling off (SCS_HIP_PSD_TOL_K = the factor, lab knob).
// nullptr (stand-alone projections, tests, the footer diagnostics): the fixed level above.
__device__ __forceinline__ double psd_offtol2(const double *tol2) { return tol2 ? *tol2 : kPsdOffTol2; }
constexpr size_t kPsdLdsBytes = (size_t)(kPsdWaves * kPsdWaveLds + 16 + 2) * sizeof(double) + 2 * kPsdMaxH * sizeof(int);

struct PsdBatch {
  const int *off;    // start of each cone's vector inside the m-vector slice
  const int *order;  // matrix order n_c
  const long *woff;  // offset (doubles) of this matrix's scratch: A, V, T, V' (NP*NP each), W (NB/2 * 16x17), rotation log, lam (NP), state (kPsdStateDoubles)
  int count;
};

__host__ __device__ inline long psd_np(long n) {  // padded order: even number of 8-blocks
  long nb = (n + kPsdB - 1) / kPsdB;
  nb = (nb + 1) & ~1L;
  if (nb < 2) nb = 2;
  return nb * kPsdB;
}
// Rotation log entry: W row-major (ld 16), so that the MFMA operand fetch W[4 kk + lk][li] of a wavefront is 64 consecutive
// doubles (4 lines per instruction; the column-major stride-17 LDS layout read from global memory costs 17)
__host__ __device__ inline int psd_log_at(int row, int col) { return col + 16 * row; }
constexpr int kPsdLogSweeps = 6;  // split mode: sweeps whose pivot rotations fit in the log of one round (round 5: 6 x 2 rounds instead of 4 x 3 — two launches fewer per projection; same rotations in the same order: same bits)
__host__ __device__ inline long psd_log_doubles(long n) {  // rotation log: sweeps x steps x pivots x 16x17
  const long np = psd_np(n), nb = np / kPsdB;
  return (long)kPsdLogSweeps * (nb - 1) * (nb / 2) * kPsdWsz;
}
// state block at the end of a matrix's scratch:
//   [0] consecutive warm-started calls (0 = V invalid)        split mode: [1] sweeps finished  [2] steps logged in this round
//   [3..5] barrier counters of k_psd_sweep_mc, one per round  [6] XCD mask of the group
//   [7] refinement stage of this call: 0 none, 1 requested by the sweep kernel (gate passed), 2 done but its test failed (sweeps resumed)
//   [8] kfro2 = |K1|_F^2 at the gate   [9] calls refined so far   [10] refinements whose a-posteriori test failed so far
//   [11] mixed-sign off-norm^2 / |A|_F^2 found by the test after the last refinement
constexpr int kPsdStateDoubles = 16;
__host__ __device__ inline long psd_scratch_doubles(long n) {
  const long np = psd_np(n);
  return 4 * np * np + (np / 16) * kPsdWsz + psd_log_doubles(n) + np + kPsdStateDoubles;
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) double lds_f64;  // explicit LDS address space for pointers hipcc would treat as generic

__device__ __forceinline__ void rr_pair(int r, int k, int N, int &p, int &q) {
  // round-robin tournament on N (even) players, round r in [0, N-1)
  if (k == 0) { p = N - 1; q = r % (N - 1); }
  else { p = (r + k) % (N - 1); q = (r - k + (N - 1)) % (N - 1); }
  if (p > q) { const int t = p; p = q; q = t; }
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Jacobi rotation (c, s) annihilating a_pq.  The ANGLE may be approximate (hardware rcp / rsq / sqrt
// estimates, ~1e-7 relative: the rotation then leaves 1e-7 |a_pq| behind, which the next sweep removes),
// but (c, s) must be orthonormal to full precision: c = rsqrt(1 + t^2) is refined by two Newton steps.
__device__ __forceinline__ void jacobi_rot(double app, double aqq, double apq, double &c, double &s) {
  // theta = (aqq - app) / (2 apq) in fp64 (one hardware reciprocal estimate), the tangent
  // t = sign(theta) / (|theta| + sqrt(theta^2 + 1)) in fp32 (|t| <= 1; an overflowing theta gives t = 0, the right limit),
  // c = (1 + t^2)^(-1/2) from the fp32 estimate by three fp64 Newton steps (1e-7 -> 1e-14 -> full precision).
  const double theta = (aqq - app) * 0.5 * __builtin_amdgcn_rcp(apq);
  const float tf = (float)theta;
  const float af = fabsf(tf);
  float t32 = __builtin_amdgcn_rcpf(af + __builtin_amdgcn_sqrtf(af * af + 1.0f));
  t32 = (af < 1e18f) ? t32 : 0.0f;  // af^2 overflows beyond ~1.8e19; t < 3e-19 there anyway
  const double t = tf >= 0.0f ? (double)t32 : -(double)t32;
  const double z = t * t + 1.;
  double y = (double)__builtin_amdgcn_rsqf((float)z);
  y = y * (1.5 - 0.5 * z * y * y);
  y = y * (1.5 - 0.5 * z * y * y);
  y = y * (1.5 - 0.5 * z * y * y);
  c = y;
  s = t * y;
}

// One wavefront works on the symmetric 16x16 pivot S = [[App, Apq], [Aqp, Aqq]] (LDS, column-major), W <- accumulated
// rotations (LDS).  Which pairs: the 64 CROSS pairs (i in block p, j in block q) in 8 rounds of 8 disjoint rotations
// (i, 8 + (i + r) mod 8) — and, in the first outer step of a sweep (`intra`), the 2 x 28 pairs INSIDE the two diagonal
// blocks first (7 rounds: a round-robin on 8 indices in each block).  Every block sits in exactly one pivot of outer step
// 0, so one outer sweep rotates every index pair of the matrix exactly once: a cyclic Jacobi method in block order.
// (Round 1 ran a full cyclic sweep of the 16x16 pivot — 15 rounds, the intra-block pairs of every block rotated again in
// each of its NB - 1 pivots.  Same number of outer sweeps to 1e-8 in tools/dbg/block_jacobi_model.py — cold 7 = 7,
// warm 3 = 3 and 2 = 2 — at 8 instead of 15 rounds per pivot: the pivot solve is the critical path of K9.)
// Lane (k = lane&7, k2 = lane>>3) owns the 2x2 block rows{p,q} x cols{p2,q2} of the round's pairs k and k2; the
// schedule is computed arithmetically (no table look-up in front of the dependent LDS reads).
// A round is ONE LDS round trip: every lane reads the three entries that define the rotations of both its
// pairs (the 8 lanes sharing a pair compute the same (c,s) redundantly — no exchange, no second barrier),
// its 2x2 block of S and its two rows of W, rotates, writes back.
constexpr int kPsdInnerSweeps = PSD_INNER;
__device__ __forceinline__ void intra8(int r, int k, int &p, int &q) {  // round r < 7 of the round-robin inside both 8-blocks
  const int kb = k & 3, base = (k >> 2) * 8;
  int a = r + kb, b = r - kb + 7;
  a = a >= 7 ? a - 7 : a;
  b = b >= 7 ? b - 7 : b;
  if (kb == 0) { a = 7; b = r; }
  p = base + min(a, b);
  q = base + max(a, b);
}
__device__ inline void wave_jacobi16(double *S, double *W, int lane, bool intra) {
  for (int e = lane; e < 256; e += 64) W[(e & 15) + kPsdWLd * (e >> 4)] = ((e & 15) == (e >> 4)) ? 1. : 0.;
  wave_sync();
  const int i0 = lane >> 3, i1 = i0 + 8;  // the two rows of W this lane rotates (columns p,q of its pair k)
  for (int sweep = 0; sweep < kPsdInnerSweeps; ++sweep) {
    if (kPsdInnerSweeps > 1) {  // a single pass needs no stopping test (two wavefront reductions, 0.6 us of a 4 us solve):
                                // entries below 1e-300 are not rotated, and the outer sweeps stop on the whole matrix
      double off = 0., tot = 0.;
      for (int e = lane; e < 256; e += 64) {
        const double a = S[(e & 15) + kPsdLd * (e >> 4)];
        tot += a * a;
        if ((e & 15) != (e >> 4)) off += a * a;
      }
      off = wave_sum(off);
      tot = wave_sum(tot);
      off = __shfl(off, 0, 64);
      tot = __shfl(tot, 0, 64);
      if (off <= 1e-26 * tot || off == 0.) break;  // (16 eps)^2 ~ 1e-29 is the rounding floor
    }
    for (int rr = intra ? 0 : 7; rr < 15; ++rr) {
      int p, q, p2, q2;
      if (rr < 7) {  // inside the diagonal blocks
        intra8(rr, lane & 7, p, q);
        intra8(rr, lane >> 3, p2, q2);
      } else {       // block p against block q
        p = lane & 7;
        q = 8 + ((p + rr - 7) & 7);
        p2 = lane >> 3;
        q2 = 8 + ((p2 + rr - 7) & 7);
      }
      const double dpp = S[p + kPsdLd * p], dqq = S[q + kPsdLd * q], dpq = S[p + kPsdLd * q];
      const double app = S[p + kPsdLd * p2], apq = S[p + kPsdLd * q2], aqp = S[q + kPsdLd * p2], aqq = S[q + kPsdLd * q2];
      const double wp0 = W[i0 + kPsdWLd * p], wq0 = W[i0 + kPsdWLd * q], wp1 = W[i1 + kPsdWLd * p], wq1 = W[i1 + kPsdWLd * q];
      // rotation of this lane's row pair k = lane & 7 (branch-free; the 8 lanes sharing k compute the same bits);
      // the column pair's rotation (k2 = lane >> 3) is lane k2's own: fetched by a wave shuffle instead of a second
      // evaluation (the 13 concurrent pivot solves of an order-200 matrix are VALU-bound, not latency-bound)
      const bool rot = fabs(dpq) > 1e-300;
      double c, s;
      jacobi_rot(dpp, dqq, rot ? dpq : 1.0, c, s);
      c = rot ? c : 1.;
      s = rot ? s : 0.;
      const double c2 = __shfl(c, lane >> 3, 64), s2 = __shfl(s, lane >> 3, 64);
      wave_sync();  // every lane has read S and W of the previous round's state
      const double t1 = c2 * app - s2 * apq, t2 = s2 * app + c2 * apq;
      const double t3 = c2 * aqp - s2 * aqq, t4 = s2 * aqp + c2 * aqq;
      S[p + kPsdLd * p2] = c * t1 - s * t3;
      S[p + kPsdLd * q2] = c * t2 - s * t4;
      S[q + kPsdLd * p2] = s * t1 + c * t3;
      S[q + kPsdLd * q2] = s * t2 + c * t4;
      W[i0 + kPsdWLd * p] = c * wp0 - s * wq0;
      W[i0 + kPsdWLd * q] = s * wp0 + c * wq0;
      W[i1 + kPsdWLd * p] = c * wp1 - s * wq1;
      W[i1 + kPsdWLd * q] = s * wp1 + c * wq1;
      wave_sync();
    }
  }
}

// index of local row/col i (0..15) of the pivot (p,q): block p for i<8, block q otherwise
__device__ __forceinline__ int pq_index(int i, int p, int q) { return (i < 8 ? p * kPsdB : q * kPsdB - 8) + i; }

// acc[j] = (row tile ti of Aop) x (row tile tj0 + j of Bop)', j < kPsdNJ: C[i][jj] = sum_k Aop[i][k] Bop[jj][k], both
// operands column-major with the contraction index along the columns, so every load runs down a column (full
// 128-byte lines).  The a-operand is shared by the kPsdNJ output tiles (1.25 loads per MFMA instead of 2).
// Tiles beyond `tjmax` are computed on a clamped tile and must be ignored by the caller.
constexpr int kPsdNJ = 4;
// FMAP: the B operand is F = Pi_+(D + E) formed on the fly from the nearly diagonal A (see the reconstruction):
// F_jk = max(d_k, 0) on the diagonal, A_jk * (f(d_j) - f(d_k)) / (d_j - d_k) elsewhere; d = lam[] (eigenvalue estimates).
__device__ __forceinline__ double psd_fmap(double a, double dj, double dk, bool diag) {
  const double hi = fmax(dj, dk), lo = fmin(dj, dk);
  const double gdd = lo > 0. ? 1. : (hi <= 0. ? 0. : hi / (hi - lo));
  return diag ? fmax(dk, 0.) : a * gdd;
}
template <bool FMAP = false>
__device__ __forceinline__ void mma_row(const double *__restrict__ Aop, const double *__restrict__ Bop, int ld, int NP, int ti,
                                        int tj0, int tjmax, int li, int lk, f64x4 (&acc)[kPsdNJ], const double *lam = nullptr) {
  const double *pa = Aop + (ti * 16 + li) + (size_t)ld * lk;
  const double *pb[kPsdNJ];
  int rowj[kPsdNJ];
  double dj[kPsdNJ];
#pragma unroll
  for (int j = 0; j < kPsdNJ; ++j) {
    acc[j] = f64x4{0., 0., 0., 0.};
    rowj[j] = min(tj0 + j, tjmax) * 16 + li;
    pb[j] = Bop + rowj[j] + (size_t)ld * lk;
    dj[j] = FMAP ? lam[rowj[j]] : 0.;
  }

  for (int k0 = 0; k0 < NP; k0 += 4) {
    const double a = pa[(size_t)ld * k0];
    double b[kPsdNJ];
#pragma unroll
    for (int j = 0; j < kPsdNJ; ++j) b[j] = pb[j][(size_t)ld * k0];
    if (FMAP) {
      const double dk = lam[k0 + lk];
#pragma unroll
      for (int j = 0; j < kPsdNJ; ++j) b[j] = psd_fmap(b[j], dj[j], dk, rowj[j] == k0 + lk);
    }
#pragma unroll
    for (int j = 0; j < kPsdNJ; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[j], acc[j], 0, 0, 0);
  }
}

// ---- wave-level GEMM tasks shared by the one-workgroup kernels (task = wave, wave + 16, ...) and the multi-workgroup
// ---- k_psd_gemm (task = global wave index): `Sw` is the calling wave's private 16x17 LDS scratch ----
// G1: Tt = Vt A  (A exactly symmetric: A[k][i] is read as A[i][k]); output tiles (tj, ti0 .. ti0+3) of Tt = (A V)'
__device__ __forceinline__ void psd_task_g1(int task, int NP, const double *A, const double *Vt, double *Tm, double *Sw, int li, int lk) {
  const int ld = NP, ntile = NP / 16;
  const int tj = task % ntile, ti0 = (task / ntile) * kPsdNJ;
  f64x4 acc[kPsdNJ];
  mma_row(Vt, A, ld, NP, tj, ti0, ntile - 1, li, lk, acc);
#pragma unroll
  for (int j = 0; j < kPsdNJ; ++j) {
    const int ti = ti0 + j;
    if (ti >= ntile) break;
    // lane holds Tt[row = tj*16 + lk + 4t][col = ti*16 + li]
#pragma unroll
    for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = acc[j][t];
    wave_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) Tm[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] = Sw[li + 17 * (lk + 4 * t)];
    wave_sync();
  }
}
// G2: A0[i][j] = sum_k Vt[i][k] Tt[j][k], lower-triangular tiles, mirrored; diagonal tiles symmetrised (average of
// the two triangles) so that A0 is exactly symmetric
__device__ __forceinline__ void psd_task_g2(int task, int NP, double *A, const double *Vt, const double *Tm, double *Sw, int li, int lk) {
  const int ld = NP, ntile = NP / 16;
  const int ti = task % ntile, tj0 = (task / ntile) * kPsdNJ;
  if (tj0 > ti) return;
  f64x4 acc[kPsdNJ];
  mma_row(Vt, Tm, ld, NP, ti, tj0, ti, li, lk, acc);
#pragma unroll
  for (int j = 0; j < kPsdNJ; ++j) {
    const int tj = tj0 + j;
    if (tj > ti) break;
    // lane holds R[row = lk + 4t][col = li] of tile (ti, tj)
    if (ti != tj) {
#pragma unroll
      for (int t = 0; t < 4; ++t) A[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] = acc[j][t];  // mirror
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = acc[j][t];
    wave_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = li, c = lk + 4 * t;  // element (r, c) of the tile = Sw[r + 17 c]
      double v = Sw[r + 17 * c];
      if (ti == tj && r != c) v = 0.5 * ((r > c ? v : Sw[c + 17 * r]) + (r > c ? Sw[c + 17 * r] : v));
      A[(ti * 16 + r) + (size_t)ld * (tj * 16 + c)] = v;
    }
    wave_sync();
  }
}
// R1: T = V F  (F symmetric, formed on the fly from the diagonalised A and lam = its diagonal)
// (FMAP = false: `A` is F itself, formed once by k_psd_fmap — the split pipeline; the divided differences cost a division each and
// every element of F is an operand of NP / 16 row tiles)
template <bool FMAP = true>
__device__ __forceinline__ void psd_task_r1(int task, int NP, const double *A, const double *V, double *Tm, const double *lam,
                                            double *Sw, int li, int lk) {
  const int ld = NP, ntile = NP / 16;
  const int ti = task % ntile, tj0 = (task / ntile) * kPsdNJ;
  f64x4 acc[kPsdNJ];
  mma_row<FMAP>(V, A, ld, NP, ti, tj0, ntile - 1, li, lk, acc, lam);
#pragma unroll
  for (int j = 0; j < kPsdNJ; ++j) {
    const int tj = tj0 + j;
    if (tj >= ntile) break;
#pragma unroll
    for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = acc[j][t];
    wave_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) Tm[(ti * 16 + li) + (size_t)ld * (tj * 16 + lk + 4 * t)] = Sw[li + 17 * (lk + 4 * t)];
    wave_sync();
  }
}
// R2: X+ = T V', lower-triangular output tiles straight into the packed vector
__device__ __forceinline__ void psd_task_r2(int task, int n, int NP, const double *Tm, const double *V, double *X, int li, int lk) {
  const int ld = NP, ntile = NP / 16;
  const double sq2 = 1.41421356237309504880;
  const int ti = task % ntile, tj0 = (task / ntile) * kPsdNJ;
  if (tj0 > ti) return;
  f64x4 acc[kPsdNJ];
  mma_row(Tm, V, ld, NP, ti, tj0, ti, li, lk, acc);
#pragma unroll
  for (int jj = 0; jj < kPsdNJ; ++jj) {
    const int tj = tj0 + jj;
    if (tj > ti) break;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = ti * 16 + lk + 4 * t, j = tj * 16 + li;
      if (i < n && j <= i) {
        const long base = (long)j * n - (long)j * (j - 1) / 2;
        X[base + (i - j)] = (i == j) ? acc[jj][t] : acc[jj][t] * sq2;
      }
    }
  }
}

// ---- X+ = V F V' with F = Pi_+(D + E) to second order in the remaining off-diagonal part E of A = D + E ----
// (one 1024-lane workgroup; Tm is overwritten by V F; Sw = this wave's 16x17 LDS scratch)
__device__ __forceinline__ void psd_reconstruct(double *X, int n, int NP, double *A, const double *V, double *Tm, double *lam,
                                                double *Sw) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int ld = NP, ntile = NP / 16;
  // The sweeps stop at ||E||_F <= 1e-8 ||A||_F, one sweep earlier than a plain V max(D,0) V' would allow
  // (its error is first order in E).  For a matrix function f applied to a nearly diagonal matrix,
  //   f(D + E)_ij = f(d_i) delta_ij + E_ij (f(d_i) - f(d_j)) / (d_i - d_j) + O(|E|^2 / gap)      (Daleckii-Krein),
  // and for f = max(., 0) the divided difference is 1 (both positive), 0 (both non-positive) or
  // hi / (hi - lo) in (0, 1] for a pair straddling zero — always well defined.  The eigenvalue estimates d_i are
  // themselves second-order accurate, so the result is good to ~|E|^2 = 1e-16 relative.
  for (int j = tid; j < NP; j += kPsdThreads) lam[j] = (j < n) ? A[j + (size_t)ld * j] : 0.;
  __syncthreads();
  const int ntask = ntile * ((ntile + kPsdNJ - 1) / kPsdNJ);
  for (int task = wave; task < ntask; task += kPsdWaves) psd_task_r1(task, NP, A, V, Tm, lam, Sw, li, lk);
  __syncthreads();
  for (int task = wave; task < ntask; task += kPsdWaves) psd_task_r2(task, n, NP, Tm, V, X, li, lk);
}

// ---- What the sweeps of the split pipeline stop on (round 5) ----
// The projection does not need the eigenvectors, only the SPLIT between the positive and the non-positive invariant subspace: for
// S = V'AV = [[S_PP, C], [C', S_NN]] (P: positive diagonal entries, N: the others) with S_PP > 0 >= S_NN and C = 0 the second-order
// reconstruction F = Pi_+(S) = [[S_PP, 0], [0, 0]] is EXACT whatever is left inside the two diagonal blocks, and for C != 0 its error
// is at most |C| (Pi_+ is 1-Lipschitz), ~|C|^2 / gap in practice.  In the warm-started steady state of ADMM S is nearly diagonal and
// the mixed-sign part is removed by ONE step of a GEMM-only refinement (Ogita-Aishima restricted to the mixed-sign pairs, with the
// same-sign blocks kept in the Sylvester operator to second order) instead of a Jacobi sweep — the latency chain of 25 outer steps:
//     K1_ij = S_ij / (d_j - d_i)  (d_i d_j mixed),   K2_ij = (S_ij + [S_off, K1]_ij) / (d_j - d_i),   Q = I + K2 + K2^2 / 2,
//     V <- V Q,   S1 = Q' S Q,   |C(S1)| = O(third order)          (k_psd_plan, k_psd_gemm<COMM / KK / T / S1>, k_psd_apply_q)
// (tools/dbg/psd_refine_model.py on the matrices config 4 really projects: |C| 1e-4 |A| -> 1e-10 .. 5e-9 |A|, error against LAPACK
// <= 1.2e-11; Q is orthogonal to |K2|^4 / 4 <= 1e-11, and V is re-orthogonalised every kPsdWarmPeriod calls as before.)
//   PSD_STOP_STRICT   the test of rounds 1-4: |off(A)|_F^2 <= tol2 |A|_F^2 (identical sums: identical decisions, identical bits)
//   PSD_STOP_GATE     ... or REFINABLE (code 2): |K1|_F^2 <= R.k2, |off|^2 <= R.off2 |A|^2 and omega <= R.omega, where
//                     omega = sum over same-sign pairs of a_ij^2 / (d_i d_j) bounds |D^-1/2 E D^-1/2|_F^2 of both diagonal blocks
//                     (< 1 => they are definite); the third-order remainder of the step is then below the strict level
//   PSD_STOP_RELAXED  after a refinement: mixed-sign off-norm^2 <= tol2 |A|^2 and omega <= R.omega_relaxed — what the
//                     reconstruction needs; a matrix that fails it goes back to the sweeps (from S1, with the refined V: nothing is lost)
enum : int { PSD_STOP_STRICT = 0, PSD_STOP_GATE = 1, PSD_STOP_RELAXED = 2 };
#if PSD_PROFILE
// tools/psd_lab.hip: 10 ns ticks of the LAST stopping test run by thread 0 of workgroup 0 — [0] diagonal [1] reciprocal table [2] the
// sums (+ K1) [3] block reductions + decision [4] calls so far [5] mode of the last call
__device__ double psd_prof_stop[8];
#define PSD_STOP_TICK(var) const long long var = (threadIdx.x == 0 && blockIdx.x == 0) ? wall_clock64() : 0
#else
#define PSD_STOP_TICK(var) do { } while (0)
#endif
struct PsdRefineCfg {
  int on;  // 0: strict sweeps only (bit-identical to the one-launch kernel)
  double k2, off2, omega, omega_relaxed;
};
__host__ inline PsdRefineCfg psd_refine_default(bool on) { return PsdRefineCfg{on ? 1 : 0, 3.6e-5, 2.25e-6, 0.05, 0.25}; }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every lane of the 1024-lane workgroup returns the code: 0 go on sweeping, 1 converged, 2 refinable.  `diag`: n doubles of LDS;
// `lead`: this workgroup records the statistics of the decision in state[].  All members of a multi-CU group evaluate the test on
// the same data in the same order: the same decision everywhere.
// K1 (PSD_STOP_GATE, lead only): the first step of the refinement, K1_ij = a_ij / (d_j - d_i) on the mixed-sign pairs and 0 elsewhere
// (padding included), is written while the gate's sums are formed — speculatively: whether the matrix IS refinable is known a
// reduction later, and a matrix that is not simply never reads it.  a_ij = a_ji exactly and the reciprocal estimate is odd in its
// argument: K1 is exactly antisymmetric.
// (ONE call site per kernel — k_psd_sweep_mc sits at its 128-register cap with spills — and inlined: as a real function it needs a dynamic stack,
//  which a cooperative launch aborts on)
template <bool AGENT>
__device__ __forceinline__ int psd_stop_test(const double *A, int ld, int n, double *diag, double *red, double *bc, double offtol2, int mode,
                                             const PsdRefineCfg &R, double *state, bool lead, double *K1 = nullptr, bool first = false) {
  const int tid = threadIdx.x;
  const bool wk = K1 != nullptr && lead && mode == PSD_STOP_GATE;
  PSD_STOP_TICK(ts0);
  if (wk) {  // rows / columns of the padding
    const int NP = ld;
    for (int e = tid; e < (NP - n) * NP; e += kPsdThreads) {
      const int j = n + e / NP, i = e % NP;  // columns n .. NP-1 whole, then the rows n .. NP-1 of the other columns
      K1[i + (size_t)ld * j] = 0.;
      if (i < n) K1[j + (size_t)ld * i] = 0.;
    }
  }
  if (mode != PSD_STOP_STRICT) {
    for (int j = tid; j < n; j += kPsdThreads) diag[j] = AGENT ? ld_agent(&A[j + (size_t)ld * j]) : A[j + (size_t)ld * j];
    __syncthreads();
  }
  double off = 0., tot = 0., mix = 0., kf = 0., om = 0.;
  PSD_STOP_TICK(ts1);
#if PSD_PROFILE
  long long ts2 = 0;
#endif
  if (mode != PSD_STOP_STRICT) {
    // Column by column (wavefront w: columns w, w + 16, ...; lanes down the rows), the LOWER triangle only — the matrix is exactly symmetric, every
    // off-diagonal term counts twice —, the same-sign terms through a table of 1 / d_i, the mixed-sign ones through ONE fp32 reciprocal estimate
    // (a gate, and K1 only enters the second-order term; the estimate is odd in its argument: K1 stays exactly antisymmetric).  Round 5: the test is
    // what a projection that needs no sweep still pays twice (before and behind the refinement); the first version (all n^2 elements, two fp64
    // reciprocals each) took 32 us of a 377 us projection per call.  (The strict test below keeps the element order of rounds 1-4 — the bits of the
    // one-launch kernel's decision.)
    const int wave = tid >> 6, lane = tid & 63;
    const bool tab = 2 * n <= kPsdWaves * kPsdWaveLds;  // 1 / d_i behind the diagonal in LDS (orders beyond 4352: reciprocals on the fly)
    if (tab) {
      for (int j = tid; j < n; j += kPsdThreads) diag[n + j] = 1. / diag[j];
      __syncthreads();
    }
#if PSD_PROFILE
    if (threadIdx.x == 0 && blockIdx.x == 0) ts2 = wall_clock64();
#endif
    // Round 6: the loads of a wavefront's trips go out EIGHT at a time, every one of them unconditional (the row index is clamped: a
    // load inside a condition makes hipcc wait for it on the spot).  A trip's load used to be consumed right behind its issue — ~32
    // dependent round trips to an L2 that may belong to another XCD, 0.5 us each: 16 us of a 27 us test that reads 160 KB, and a
    // projection of the steady state pays for two such tests (tools/psd_lab.hip -DPSD_PROFILE=1).  The per-lane order of the
    // accumulation is unchanged (column by column, trip by trip): the same sums, the same decisions.
    constexpr int U = 8;
    int cj = wave, ck = 0;  // cursor: column, trip inside the column (the same in every lane of the wavefront)
    while (cj < n) {
      double av[U];
      int ei[U], ej[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = cj < n;
        const int j = live ? cj : 0, i = live ? cj + 64 * ck + lane : n;
        ej[u] = j;
        ei[u] = i;
        const double *pa = &A[min(i, n - 1) + (size_t)ld * j];
        av[u] = AGENT ? ld_agent(pa) : *pa;
        ck += live ? 1 : 0;
        const bool next = live && cj + 64 * ck >= n;
        cj += next ? kPsdWaves : 0;
        ck = next ? 0 : ck;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = ei[u], j = ej[u];
        if (i >= n) continue;
        const double dj = diag[j];
        const bool pj = dj > 0.;
        const double ij = tab ? diag[n + j] : 1. / dj;
        const double a = av[u];
        const double a2 = a * a;
        if (i == j) {
          tot += a2;
          if (wk) K1[i + (size_t)ld * j] = 0.;
          continue;
        }
        tot += 2. * a2;
        off += 2. * a2;
        const double di = diag[i];
        double k1 = 0.;  // K1[i][j] = a / (d_j - d_i)
        if ((di > 0.) != pj) {
          k1 = a * (double)__builtin_amdgcn_rcpf((float)(dj - di));
          mix += 2. * a2;
          kf += 2. * k1 * k1;
        } else if (a2 > 0.) {
          om += 2. * a2 * ((tab ? diag[n + i] : 1. / di) * ij);  // same sign: positive; a zero diagonal entry under a nonzero row: inf, no refinement
        }
        if (wk) {
          K1[i + (size_t)ld * j] = k1;
          K1[j + (size_t)ld * i] = -k1;
        }
      }
    }
  } else
  for (int e = tid; e < n * n; e += kPsdThreads) {
    const int i = e % n, j = e / n;
    const double a = AGENT ? ld_agent(&A[i + (size_t)ld * j]) : A[i + (size_t)ld * j];
    const double a2 = a * a;
    tot += a2;
    if (i != j) {
      off += a2;
      if (mode != PSD_STOP_STRICT) {
        const double di = diag[i], dj = diag[j];
        double k1 = 0.;
        if ((di > 0.) != (dj > 0.)) {
          const double r = __builtin_amdgcn_rcp(dj - di);  // (a hardware estimate is enough for a gate, and K1 only enters the second-order term)
          k1 = a * r;
          mix += a2;
          kf += k1 * k1;
        } else if (a2 > 0.) {
          om += a2 * __builtin_amdgcn_rcp(di * dj);  // same sign: positive; a zero diagonal entry under a nonzero row: inf, no refinement
        }
        if (wk) K1[i + (size_t)ld * j] = k1;
      }
    } else if (wk) {
      K1[i + (size_t)ld * j] = 0.;
    }
  }
#if PSD_PROFILE
  __syncthreads();  // (profile builds: the slowest wavefront ends the sums)
#endif
  PSD_STOP_TICK(ts3);
  if (mode != PSD_STOP_STRICT && 2 * n + 5 * kPsdWaves <= kPsdWaves * kPsdWaveLds) {
    // the five sums in ONE pass (round 6: five block_sum calls were ten barriers, 6.5 us): each value through the same shuffle tree and
    // the same 16-term sum in wavefront order as block_sum — the same bits —, the partials behind the reciprocal table in LDS
    double v5[5] = {off, tot, mix, kf, om};
#pragma unroll
    for (int q = 0; q < 5; ++q) v5[q] = wave_sum(v5[q]);
    double *sm5 = diag + 2 * n;
    if ((tid & 63) == 0)
      for (int q = 0; q < 5; ++q) sm5[q * kPsdWaves + (tid >> 6)] = v5[q];
    __syncthreads();
    if (tid == 0) {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        double r = 0.;
#pragma unroll
        for (int i = 0; i < kPsdWaves; ++i) r += sm5[q * kPsdWaves + i];
        v5[q] = r;
      }
      off = v5[0]; tot = v5[1]; mix = v5[2]; kf = v5[3]; om = v5[4];
    }
  } else {
  off = block_sum<kPsdThreads>(off, red);
  tot = block_sum<kPsdThreads>(tot, red);
  if (mode != PSD_STOP_STRICT) {
    mix = block_sum<kPsdThreads>(mix, red);
    kf = block_sum<kPsdThreads>(kf, red);
    om = block_sum<kPsdThreads>(om, red);
  }
  }
  if (tid == 0) {
    int code = (off <= offtol2 * tot || off == 0.) ? 1 : 0;
    if (!code && mode == PSD_STOP_GATE && kf <= R.k2 && off <= R.off2 * tot && om <= R.omega) code = 2;
    if (!code && mode == PSD_STOP_RELAXED && mix <= offtol2 * tot && om <= R.omega_relaxed) code = 1;
    if (lead && code == 2) state[8] = kf;
    if (lead && mode == PSD_STOP_GATE && first) { state[12] = kf; state[13] = tot > 0. ? off / tot : 0.; state[14] = om; }  // (diagnostics: what the matrix looked like as it arrived)
    if (lead && mode == PSD_STOP_RELAXED) state[11] = tot > 0. ? mix / tot : 0.;
    bc[0] = (double)code;
  }
  __syncthreads();
  const int code = (int)bc[0];
  __syncthreads();
#if PSD_PROFILE
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long ts4 = wall_clock64();
    psd_prof_stop[0] = (double)(ts1 - ts0);
    psd_prof_stop[1] = (double)((ts2 ? ts2 : ts1) - ts1);
    psd_prof_stop[2] = (double)(ts3 - (ts2 ? ts2 : ts1));
    psd_prof_stop[3] = (double)(ts4 - ts3);
    psd_prof_stop[4] += 1.;
    psd_prof_stop[5] = (double)mode;
  }
#endif
  return code;
}

// MODE 0: the whole projection in one launch (one workgroup = one CU per matrix; right when the batch fills the GPU).
// Split mode for small batches of large matrices (config 4: 50 matrices on 256 CUs), everything that parallelises
// beyond one CU per matrix in its own multi-workgroup launch:
//   k_psd_front: unpack, V = I / V', counters (many workgroups per matrix); MODE 3: the periodic re-orthogonalisation of V
//           (one workgroup per matrix; returns at once on the other calls).  MODE 2 = both in one workgroup (kept for A/B)
//   k_psd_gemm<G1>, <G2>: warm start A0 = V'AV                                        7 workgroups / matrix at order 200
//   MODE 1  sweeps: diagonalise A (pivots + A updates), LOG every pivot's 16x16 rotation; a round logs at most
//           kPsdLogSweeps sweeps, the host enqueues [sweep, apply] rounds back to back and later rounds return at
//           once when the matrix has already converged (state[1])                     1 workgroup / matrix
//   k_psd_apply_v: V <- V W_1 W_2 ... per 16-row strip (the V update is 2/3 of the update work)   13 / matrix
//   k_psd_fmap: F = Pi_+(D + E) element by element; k_psd_gemm<R1>, <R2>: X+ = V F V'   7 / matrix (2 x 4 tiles per wavefront)
// Same rotations, same MFMA sequences: bit-identical to MODE 0.
// MODE 1, `R.on`: rounds before the refinement stage stop on PSD_STOP_GATE, the round behind it (`post`) re-tests a refined matrix with
// PSD_STOP_RELAXED — the same decisions as k_psd_sweep_mc, bit for bit.
template <int MODE>
__global__ __launch_bounds__(kPsdThreads) void k_proj_psd(double *x, PsdBatch B, double *scratch, int allow_warm, int round, const int *stall,
                                                          const double *tol2, PsdRefineCfg R, int post) {
  SCS_STALL_GUARD(stall);
  const double offtol2 = psd_offtol2(tol2);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // LDS: per wave S / transpose scratch (16x17) + W (16x17) doubles | red[16] | bc[2] | outer schedule (2*kPsdMaxH ints)
  double *lds = reinterpret_cast<double *>(smem_raw);
  double *red = lds + kPsdWaves * kPsdWaveLds;
  double *bc = red + 16;
  int *osch = reinterpret_cast<int *>(bc + 2);
  const int cidx = blockIdx.x;
  const int n = B.order[cidx];
  double *X = x + B.off[cidx];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (n == 0) return;
  if (n == 1) {
    if (tid == 0) X[0] = fmax(X[0], 0.);
    return;
  }
  const int NP = (int)psd_np(n), NB = NP / kPsdB, H = NB / 2, ld = NP;
  double *A = scratch + B.woff[cidx];
  double *V = A + (size_t)NP * NP;
  double *Tm = V + (size_t)NP * NP;  // scaled eigenvectors for the reconstruction / temp of the warm start
  double *Vt = Tm + (size_t)NP * NP;  // V' for the warm start's GEMMs (every operand load then runs down a column)
  double *Wg = Vt + (size_t)NP * NP;  // H pivots' rotation blocks, 16x17 doubles each (used when H > 16)
  double *Wlog = Wg + (size_t)H * kPsdWsz;  // split mode: (step, pivot) -> 16x17 rotation, steps of this round
  double *lam = Wlog + psd_log_doubles(n);
  double *state = lam + NP;           // [0] consecutive warm-started calls (0 = V invalid); split mode: [1] converged, [2] steps logged
  const double isq2 = 0.70710678118654752440, sq2 = 1.41421356237309504880;
#if PSD_PROFILE
  double prof[8] = {0., 0., 0., 0., 0., 0., 0., 0.};  // [1] unpack [2] warm GEMMs [3] pivots [4] updates [5] norms+schedule [6] reconstruct [7] sweeps
#endif
  PSD_TICK(t_begin);
  const bool w_in_lds = H <= kPsdWaves;  // pivot k is solved by wave k and its W stays in that wave's LDS

  // Warm start.  Inside ADMM the matrix to project moves little between iterations, so the eigenvectors
  // of the previous call almost diagonalise it: start from A0 = V' A V (two MFMA GEMMs, ~1/4 sweep) and
  // the Jacobi iteration converges in 1-3 sweeps instead of ~9.  Every kPsdWarmPeriod calls V gets one
  // Newton-Schulz step V <- V (3I - V'V) / 2 (three GEMMs, ~0.4 ms at order 200) that squares its distance
  // from orthogonality, so the rounding drift of the accumulated rotations (each sweep multiplies ~n^2/2 of
  // them in) stays at machine precision for arbitrarily long solves.
  const bool warm = allow_warm && state[0] >= 1.;
  const bool reorth = warm && ((long)state[0] % kPsdWarmPeriod) == 0;
  const bool resumed = MODE == 1;  // split mode: the front and the warm-start GEMMs ran in their own launches
  const bool refined = MODE == 1 && R.on && post && state[7] == 1.;  // this call's refinement stage ran: re-test what it left
  const bool finished = MODE == 1 && round > 0 && state[1] != 0. && !refined;
  __syncthreads();  // everyone has read state[]
  if (MODE == 1) {
    if (tid == 0) {
      state[2] = 0.;  // steps logged in this round
      if (round == 0) state[1] = 0.;
    }
    if (finished) return;  // converged in an earlier round: nothing to log, k_psd_apply_v has nothing to do
  }
  double *Sw = lds + wave * kPsdWaveLds, *Ww = Sw + kPsdWsz;
  const int li = lane & 15, lk = lane >> 4;
  const int nblk = H * (H + 1) / 2, ntile = NP / 16;
  if (MODE == 3 && !reorth) return;  // k_psd_front did the unpacking and V': only the periodic re-orthogonalisation is left here

  if (!resumed) {
  // ---- unpack (lower tri, col-major, off-diag / sqrt2), zero padding; V = I when cold; inner schedule ----
  if (MODE != 3) {
  for (int e = tid; e < NP * NP; e += kPsdThreads) {
    const int i = e % NP, j = e / NP;
    A[e] = 0.;
    if (!warm) V[e] = (i == j) ? 1. : 0.;
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += kPsdThreads) {  // one flat pass: independent loads, i runs down packed column j
    const int j = e / n, i = e - j * n;
    if (i < j) continue;
    const long base = (long)j * n - (long)j * (j - 1) / 2;  // start of packed column j
    double v = X[base + (i - j)];
    if (i != j) v *= isq2;
    A[i + (size_t)ld * j] = v;
    A[j + (size_t)ld * i] = v;
  }
  __syncthreads();
  }  // MODE != 3

  PSD_TICK(t_unpacked);
  PSD_ACC(1, t_begin, t_unpacked);

  if (warm) {
    // A0 = V' A V as two GEMMs whose operand loads all run down columns (li contiguous: full 128-byte lines):
    //   Vt = V'  (16x16 tiles through the 16x17 LDS transpose)
    //   Tt = Vt A        (A is exactly symmetric: A[k][i] is read as A[i][k])      Tt = (A V)'
    //   A0[i][j] = sum_k Vt[i][k] Tt[j][k]
    for (int tile = wave; tile < ntile * ntile; tile += kPsdWaves) {
      const int ti = tile % ntile, tj = tile / ntile;
#pragma unroll
      for (int t = 0; t < 4; ++t) Sw[li + 17 * (lk + 4 * t)] = V[(ti * 16 + li) + (size_t)ld * (tj * 16 + lk + 4 * t)];
      wave_sync();
#pragma unroll
      for (int t = 0; t < 4; ++t) Vt[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] = Sw[(lk + 4 * t) + 17 * li];
      wave_sync();
    }
    __syncthreads();
    const int ngrp = (ntile + kPsdNJ - 1) / kPsdNJ;
    if (reorth) {
      // Tm = (3I - V'V) / 2   (V'V[i][j] = sum_k Vt[i][k] Vt[j][k]; symmetric, so the C layout is stored as its transpose)
      for (int task = wave; task < ntile * ngrp; task += kPsdWaves) {
        const int ti = task % ntile, tj0 = (task / ntile) * kPsdNJ;
        f64x4 acc[kPsdNJ];
        mma_row(Vt, Vt, ld, NP, ti, tj0, ntile - 1, li, lk, acc);
#pragma unroll
        for (int j = 0; j < kPsdNJ; ++j) {
          const int tj = tj0 + j;
          if (tj >= ntile) break;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = ti * 16 + lk + 4 * t, col = tj * 16 + li;
            Tm[col + (size_t)ld * row] = (row == col ? 1.5 : 0.) - 0.5 * acc[j][t];
          }
        }
      }
      __syncthreads();
      // Vt (as a plain buffer) = V Tm, then V <- it and Vt <- V' again
      for (int task = wave; task < ntile * ngrp; task += kPsdWaves) {
        const int ti = task % ntile, tj0 = (task / ntile) * kPsdNJ;
        f64x4 acc[kPsdNJ];
        mma_row(V, Tm, ld, NP, ti, tj0, ntile - 1, li, lk, acc);
#pragma unroll
        for (int j = 0; j < kPsdNJ; ++j) {
          const int tj = tj0 + j;
          if (tj >= ntile) break;
#pragma unroll
          for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = acc[j][t];
          wave_sync();
#pragma unroll
          for (int t = 0; t < 4; ++t) Vt[(ti * 16 + li) + (size_t)ld * (tj * 16 + lk + 4 * t)] = Sw[li + 17 * (lk + 4 * t)];
          wave_sync();
        }
      }
      __syncthreads();
      for (int e = tid; e < NP * NP; e += kPsdThreads) V[e] = Vt[e];
      __syncthreads();
      for (int tile = wave; tile < ntile * ntile; tile += kPsdWaves) {
        const int ti = tile % ntile, tj = tile / ntile;
#pragma unroll
        for (int t = 0; t < 4; ++t) Sw[li + 17 * (lk + 4 * t)] = V[(ti * 16 + li) + (size_t)ld * (tj * 16 + lk + 4 * t)];
        wave_sync();
#pragma unroll
        for (int t = 0; t < 4; ++t) Tm[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] = Sw[(lk + 4 * t) + 17 * li];
        wave_sync();
      }
      __syncthreads();
      for (int e = tid; e < NP * NP; e += kPsdThreads) Vt[e] = Tm[e];
      __syncthreads();
    }
    if (MODE != 2 && MODE != 3) {
      for (int task = wave; task < ntile * ngrp; task += kPsdWaves) psd_task_g1(task, NP, A, Vt, Tm, Sw, li, lk);
      __syncthreads();
      for (int task = wave; task < ntile * ngrp; task += kPsdWaves) psd_task_g2(task, NP, A, Vt, Tm, Sw, li, lk);
      __syncthreads();
    }
  }

  PSD_TICK(t_warmed);
  PSD_ACC(2, t_unpacked, t_warmed);
  }  // !resumed
  if (MODE == 3) return;
  if (MODE == 2) {
    if (tid == 0) state[3] = state[4] = state[5] = state[6] = state[7] = 0.;  // barrier counters of k_psd_sweep_mc, one per round; refinement flag
    return;
  }
  int nlog = 0;  // split mode: steps logged in this round
  int stop_mode = (MODE == 1 && R.on) ? (refined ? PSD_STOP_RELAXED : (post ? PSD_STOP_STRICT : PSD_STOP_GATE)) : PSD_STOP_STRICT;
  for (int sweep = 0; sweep < kPsdMaxSweeps; ++sweep) {
    PSD_TICK(t_sw0);
    // relative off-norm 1e-8: the reconstruction below is second-order accurate in what is left
    const int code = psd_stop_test<false>(A, ld, n, lds, red, bc, offtol2, stop_mode, R, state, true, Vt, round == 0 && sweep == 0);
    PSD_TICK(t_sw1);
    PSD_ACC(5, t_sw0, t_sw1);
    if (code != 0) {
      if (MODE == 1 && tid == 0) {
        state[1] = 1.;
        if (code == 2) state[7] = 1.;  // the refinement stage takes it from here
      }
      if (MODE == 1 && code == 2)
        for (int j = tid; j < NP; j += kPsdThreads) lam[j] = j < n ? lds[j] : 0.;  // the diagonal the gate saw (k_psd_plan)
      break;
    }
    if (MODE == 1 && refined && sweep == 0 && tid == 0) {  // the refinement left too much: back to the sweeps
      state[1] = 0.;
      state[7] = 2.;
      state[10] += 1.;
    }
    if (MODE == 1 && sweep >= kPsdLogSweeps) break;  // log full: the next round continues
#if PSD_PROFILE
    prof[7] += 1.;
#endif

    for (int r = 0; r < NB - 1; ++r) {
      PSD_TICK(t_s0);
      // outer schedule of this step -> LDS (block pairs p < q)
      for (int k = tid; k < H; k += kPsdThreads) {
        int p, q;
        rr_pair(r, k, NB, p, q);
        if (k < kPsdMaxH) { osch[2 * k] = p; osch[2 * k + 1] = q; }
      }
      __syncthreads();
      PSD_TICK(t_s1);
      PSD_ACC(5, t_s0, t_s1);
      // ---------------- phase 1: one wavefront per pivot ----------------
      for (int k = wave; k < H; k += kPsdWaves) {
        const int p = osch[2 * k], q = osch[2 * k + 1];
        for (int e = lane; e < 256; e += 64) {
          const int i = e & 15, j = e >> 4;
          Sw[i + kPsdLd * j] = A[pq_index(i, p, q) + (size_t)ld * pq_index(j, p, q)];
        }
        wave_sync();
        wave_jacobi16(Sw, Ww, lane, r == 0);
        if (!w_in_lds)
          for (int e = lane; e < kPsdWsz; e += 64) Wg[(size_t)k * kPsdWsz + e] = Ww[e];
        if (MODE == 1)
          for (int e = lane; e < 256; e += 64) Wlog[((size_t)nlog * H + k) * kPsdWsz + e] = Ww[(e >> 4) + kPsdWLd * (e & 15)];
      }
      __syncthreads();
      PSD_TICK(t_s2);
      PSD_ACC(3, t_s1, t_s2);
      // ---------------- phase 2: A <- W' A W over block pairs k <= k' and V <- V W over (row tile, pivot) ----------------
      // One task list per step: nblk A tasks, then H * ntile V tasks; wave w takes tasks w, w + 16, ...
      // Software-pipelined: the four global loads of the wave's NEXT task are in flight while the current one
      // runs its MFMAs and stores (all waves reach the same phase at the same time, so without this the
      // matrix cores idle during the loads and the memory path idles during the MFMAs).
      // (instantiated twice — rotations in LDS / in global memory — so that each instruction stream has ONE kind of operand
      //  fetch and exact wait counts)
      auto phase2 = [&](auto w_lds_tag) {
        constexpr bool kWLds = decltype(w_lds_tag)::value;
        const int ntask = MODE == 1 ? nblk : nblk + H * ntile;  // split mode: the V tasks run in k_psd_apply_v
        auto decode = [&](int task, int &k, int &k2) {  // A task: (k, k2 >= k); V task: (k, row tile) with k2 = -1 - tile
          if (task < nblk) {
            int kk = 0, rem = task;
            while (rem >= H - kk) { rem -= H - kk; ++kk; }
            k = kk;
            k2 = kk + rem;
          } else {
            const int t = task - nblk;
            k = t % H;
            k2 = -1 - t / H;
          }
        };
        auto load = [&](int k, int k2, double (&av)[4]) {
          const int p = osch[2 * k], q = osch[2 * k + 1];
          if (k2 >= 0) {
            const int p2 = osch[2 * k2], q2 = osch[2 * k2 + 1];
            const int row = pq_index(li, p, q);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) av[kk] = A[row + (size_t)ld * pq_index(4 * kk + lk, p2, q2)];
          } else {
            const int rt = -1 - k2;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) av[kk] = V[(rt * 16 + li) + (size_t)ld * pq_index(4 * kk + lk, p, q)];
          }
        };
        auto run = [&](int k, int k2, const double (&av)[4]) {
          const int p = osch[2 * k], q = osch[2 * k + 1];
          // W operands into registers through two typed paths (LDS / global): a pointer that may be either compiles to FLAT
          // loads, and a flat load makes every later wait a full s_waitcnt vmcnt(0) lgkmcnt(0) — no prefetch survives that
          auto ld_w = [&](int kp, double (&w)[4]) {
            if (kWLds) {
              const lds_f64 *pw = (const lds_f64 *)(lds + kp * kPsdWaveLds + kPsdWsz);  // ds_read, not flat_load
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) w[kk] = pw[(4 * kk + lk) + kPsdWLd * li];
            } else {
              const double *pw = Wg + (size_t)kp * kPsdWsz;
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) w[kk] = pw[(4 * kk + lk) + kPsdWLd * li];
            }
          };
          double w1[4];
          ld_w(k, w1);
          if (k2 >= 0) {
            const int p2 = osch[2 * k2], q2 = osch[2 * k2 + 1];
            double w2[4];
            ld_w(k2, w2);
            f64x4 T = {0., 0., 0., 0.};  // T = Bm * W2
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              T = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], w2[kk], T, 0, 0, 0);
            f64x4 Rr = {0., 0., 0., 0.};  // R = W1' * T  (B operand of k-step t is T[t])
#pragma unroll
            for (int t = 0; t < 4; ++t)
              Rr = __builtin_amdgcn_mfma_f64_16x16x4f64(w1[t], T[t], Rr, 0, 0, 0);
            // Stores.  Lane holds R[row = lk + 4t][col = li].  The mirror block (k2,k) = R' is written straight
            // from this layout (li runs down a column: full 128-byte lines).  The direct block goes through a
            // 16x17 LDS transpose in the wave's private scratch so that li runs down its columns as well.
            const int gi = pq_index(li, p2, q2);
            if (k != k2) {
#pragma unroll
              for (int t = 0; t < 4; ++t) A[gi + (size_t)ld * pq_index(lk + 4 * t, p, q)] = Rr[t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = Rr[t];
            wave_sync();
            const int gr = pq_index(li, p, q);
#pragma unroll
            for (int t = 0; t < 4; ++t) A[gr + (size_t)ld * pq_index(lk + 4 * t, p2, q2)] = Sw[li + 17 * (lk + 4 * t)];
            wave_sync();
          } else {
            const int rt = -1 - k2;
            f64x4 T = {0., 0., 0., 0.};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              T = __builtin_amdgcn_mfma_f64_16x16x4f64(w1[kk], av[kk], T, 0, 0, 0);  // (V_blk W)'
            // lane holds (V_blk W)[row = li][col = lk + 4t]: li runs down a column -> full-line stores
#pragma unroll
            for (int t = 0; t < 4; ++t) V[(rt * 16 + li) + (size_t)ld * pq_index(lk + 4 * t, p, q)] = T[t];
          }
        };
        // Two register buffers, the next task's loads issued before the current task runs.  The loads sit in straight-line code
        // between uniform branches (never inside a per-task condition): with `if (next exists) load` hipcc waits for ALL
        // outstanding loads (s_waitcnt vmcnt(0)) before the current task's first MFMA and the prefetch hides nothing.
        {
          double bufa[4], bufb[4];
          int ka = 0, k2a = 0, kb = 0, k2b = 0;
          int t = wave;
          if (t < ntask) {
            decode(t, ka, k2a);
            load(ka, k2a, bufa);
            while (true) {
              if (t + kPsdWaves >= ntask) { run(ka, k2a, bufa); break; }
              decode(t + kPsdWaves, kb, k2b);
              load(kb, k2b, bufb);
              run(ka, k2a, bufa);
              t += kPsdWaves;
              if (t + kPsdWaves >= ntask) { run(kb, k2b, bufb); break; }
              decode(t + kPsdWaves, ka, k2a);
              load(ka, k2a, bufa);
              run(kb, k2b, bufb);
              t += kPsdWaves;
            }
          }
        }
      };
      if (w_in_lds) phase2(std::true_type{});
      else phase2(std::false_type{});
      __syncthreads();
      PSD_TICK(t_s3);
      PSD_ACC(4, t_s2, t_s3);
      ++nlog;
    }
  }
  PSD_TICK(t_swept);
  if (MODE == 1) {
    if (tid == 0) state[2] = (double)nlog;
    return;
  }

  if (tid == 0) state[0] = warm ? state[0] + 1. : 1.;
  psd_reconstruct(X, n, NP, A, V, Tm, lam, Sw);
#if PSD_PROFILE
  __syncthreads();
  PSD_TICK(t_end);
  PSD_ACC(6, t_swept, t_end);
  if (tid == 0)
    for (int i = 1; i < 8; ++i) state[i] = prof[i];
#endif
}

// ---------------------------------------------------------------------------
// Split mode, front on many CUs: unpack the packed vector into A (one pass: every entry of the padded square is computed
// from the packed lower triangle, the same value for both triangles), V = I on a cold start, V' for the warm-start GEMMs,
// the barrier counters of k_psd_sweep_mc.  One workgroup per matrix (MODE 2) needed 65 us at 50 x order 200 on 50 CUs.
// The periodic re-orthogonalisation of V stays with the one-workgroup kernel (MODE 3, which returns at once otherwise):
// on those calls V' is formed there, after V has changed.
// ---------------------------------------------------------------------------
// Multi-workgroup launches of the split pipeline: a 1-D grid of psd_xcd_grid(per, count) workgroups, `per` of them for every matrix.
// Workgroup ids are dealt round-robin to the 8 XCDs, so (id & 7) is the XCD and the matrices are dealt to the XCDs the same way:
// everything that works on one matrix shares ONE L2 (round 5: with a (per, count) grid the 14 GEMM workgroups of a matrix were spread
// over all eight, every L2 pulled the operands of all 50 matrices, 35 MB through 4 MB, and the operand fetches ran at HBM latency).
struct PsdWg { int cidx, bx, per; bool ok; };
__device__ __forceinline__ PsdWg psd_wg(int count) {
  const int per = (int)gridDim.x / (8 * ((count + 7) / 8));
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  PsdWg w;
  w.per = per;
  w.cidx = (slot / per) * 8 + xcd;
  w.bx = slot % per;
  w.ok = w.cidx < count;
  return w;
}
__host__ inline unsigned psd_xcd_grid(int per, int count) { return 8u * (unsigned)per * (unsigned)((count + 7) / 8); }

constexpr int kPsdFrontThreads = 256;
__global__ __launch_bounds__(kPsdFrontThreads) void k_psd_front(const double *x, PsdBatch B, double *scratch, int allow_warm, const int *stall) {
  SCS_STALL_GUARD(stall);
  const PsdWg wg = psd_wg(B.count);
  if (!wg.ok) return;
  __shared__ double Sws[kPsdFrontThreads / 64][16 * 17];
  const int cidx = wg.cidx;
  const int n = B.order[cidx];
  if (n < 2) return;  // orders 0 and 1: the MODE 3 launch behind this one
  const double *X = x + B.off[cidx];
  const int NP = (int)psd_np(n), H = NP / kPsdB / 2, ld = NP, ntile = NP / 16;
  double *A = scratch + B.woff[cidx];
  double *V = A + (size_t)NP * NP;
  double *Vt = V + 2 * (size_t)NP * NP;
  double *state = Vt + (size_t)NP * NP + (size_t)H * kPsdWsz + psd_log_doubles(n) + NP;
  const bool warm = allow_warm && state[0] >= 1.;
  const bool reorth = warm && ((long)state[0] % kPsdWarmPeriod) == 0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const double isq2 = 0.70710678118654752440;
  for (long e = (long)wg.bx * kPsdFrontThreads + tid; e < (long)NP * NP; e += (long)wg.per * kPsdFrontThreads) {
    const int i = (int)(e % NP), j = (int)(e / NP);
    double v = 0.;
    if (i < n && j < n) {
      const int lo = min(i, j), hi = max(i, j);  // packed column lo, row hi
      v = X[(long)lo * n - (long)lo * (lo - 1) / 2 + (hi - lo)];
      if (i != j) v *= isq2;
    }
    A[e] = v;
    if (!warm) V[e] = (i == j) ? 1. : 0.;
  }
  if (warm && !reorth) {
    double *Sw = Sws[wave];
    for (int tile = wg.bx * (kPsdFrontThreads / 64) + wave; tile < ntile * ntile; tile += wg.per * (kPsdFrontThreads / 64)) {
      const int ti = tile % ntile, tj = tile / ntile;
#pragma unroll
      for (int t = 0; t < 4; ++t) Sw[li + 17 * (lk + 4 * t)] = V[(ti * 16 + li) + (size_t)ld * (tj * 16 + lk + 4 * t)];
      wave_sync();
#pragma unroll
      for (int t = 0; t < 4; ++t) Vt[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] = Sw[(lk + 4 * t) + 17 * li];
      wave_sync();
    }
  }
  if (wg.bx == 0 && tid == 0) state[3] = state[4] = state[5] = state[6] = state[7] = 0.;  // barrier counters of k_psd_sweep_mc, refinement flag
}

// ---------------------------------------------------------------------------
// Split mode, sweeps of ONE matrix over G workgroups (G CUs): same rotations and MFMA sequences as MODE 1 — bit-identical
// A, rotation log and state — but the 16x16 pivot solves of a step (VALU-bound: 13 wavefronts on the 4 SIMDs of one CU
// at order 200, 17 us per step) run one per SIMD on G CUs and the A-update tasks over 16 G wavefronts.
//   step = [pivot k -> workgroup k % G, wave k / G: load, solve, W -> log]  barrier  [A tasks, W from the log]  barrier
// The barrier is a monotonic counter per (matrix, round) in state[3 + round] (zeroed by the front kernel); a store is
// acknowledged (s_waitcnt vmcnt(0)) before its workgroup arrives.  Data exchange (tools/xcd_coherence_lab.hip):
//   * the G workgroups of a matrix get workgroup ids of the same residue mod 8, which the dispatcher deals to the same
//     XCD: one shared L2.  Plain stores (write-through L1, acknowledged by the L2) + agent-scope (sc1) loads (L1 bypassed,
//     L2 hit) are coherent there: 0 stale values in 5e8, 4.0 us per 32 KB ping-pong;
//   * every member publishes its HW_REG_XCC_ID first; a group that is NOT on one XCD (other partition modes, a
//     different dispatcher) falls back to sc1 (write-through to memory) stores, which are coherent across XCDs
//     (6.2 us per ping-pong) — plain stores would be 100 % stale there.
// Co-residency of every spinning workgroup: the grid is sized to one workgroup per CU of the device (work.hpp psd_mc_members), and
// inside a process the spinning launches of different workspaces / streams of a device are CHAINED (work.hpp SpinChain: each waits
// for the event behind the previous one), so two such grids never share the device; hipLaunchCooperativeKernel (SCS_HIP_PSD_COOP=1)
// makes it the runtime's guarantee at ~0.1-2 ms per launch.  What is left — another PROCESS on the same GPU — is caught by the spin
// budget: error flag, every barrier of the launch opens, the host restarts the solve with one workgroup per matrix.
// ---------------------------------------------------------------------------
constexpr int kPsdMcMaxG = 8;
__device__ __forceinline__ unsigned psd_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__host__ __device__ inline int psd_mc_grid(int count, int G) { return 8 * G * ((count + 7) / 8); }

constexpr size_t kPsdMcLdsBytes = kPsdLdsBytes + 4 * kPsdMaxH * sizeof(int);  // + next step's schedule, block -> (pair, position) map
// look_ahead (and H <= 8 G): ONE barrier per step.  While the other wavefronts run the A tasks of step t (reading A_t, writing
// A_{t+1} into the second buffer — Tm is free during the sweeps), the wavefront that owns pivot k' of step t+1 computes the
// three tasks that produce its 16x16 block — (a,a), (b,b), (a,b) for the pairs a, b that hold its two 8-blocks in step t —
// itself, keeps the results in registers, assembles the block in LDS, solves it and logs W_{t+1}: the 8 us pivot solve
// disappears behind the 6 us of A tasks.  Same MFMA sequences on the same inputs: the same bits.
__global__ __launch_bounds__(kPsdThreads) void k_psd_sweep_mc(PsdBatch B, double *scratch, int round, int G, int look_ahead, int *err,
                                                              const int *stall, const double *tol2, PsdRefineCfg R, int post, long spin_budget) {
  SCS_STALL_GUARD(stall);
  const double offtol2 = psd_offtol2(tol2);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *lds = reinterpret_cast<double *>(smem_raw);
  double *red = lds + kPsdWaves * kPsdWaveLds;
  double *bc = red + 16;
  int *osch = reinterpret_cast<int *>(bc + 2);  // block pairs (p < q) of the current step
  int *osch_n = osch + 2 * kPsdMaxH;            // ... of the next step
  int *where = osch_n + 2 * kPsdMaxH;           // block -> 2 * pair + position (0: p, 1: q) in the current step
  // workgroup id -> (matrix, member): ids are dealt round-robin to the 8 XCDs
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cidx = (slot / G) * 8 + xcd, g = slot % G;
  if (cidx >= B.count) return;
  const int n = B.order[cidx];
  if (n < 2) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int NP = (int)psd_np(n), NB = NP / kPsdB, H = NB / 2, ld = NP;
  double *A = scratch + B.woff[cidx];
  double *Wlog = A + 4 * (size_t)NP * NP + (size_t)H * kPsdWsz;
  double *lam = Wlog + psd_log_doubles(n);
  double *state = lam + NP;
  unsigned *bar = reinterpret_cast<unsigned *>(state + 3 + round);
  const bool refined = R.on && post && state[7] == 1.;  // this call's refinement stage ran: re-test what it left
  const bool finished = round > 0 && state[1] != 0. && !refined;  // written by the previous round's launch
  __syncthreads();  // everyone has read state[]
  if (g == 0 && tid == 0) {
    state[2] = 0.;
    if (round == 0) state[1] = 0.;
  }
  if (finished) return;
  // The first stopping test comes before anything spins: a matrix with nothing to do (refinable as it arrives — the steady state of
  // ADMM — or refined to the relaxed level) leaves without a barrier.  Every member decides on the same data: the same decision.
  const int stop_mode = R.on ? (refined ? PSD_STOP_RELAXED : (post ? PSD_STOP_STRICT : PSD_STOP_GATE)) : PSD_STOP_STRICT;
  double *Vt = A + 3 * (size_t)NP * NP;  // free between the warm-start GEMMs and k_psd_fmap: K1 / Q' of the refinement stage
  double *Sw = lds + wave * kPsdWaveLds, *Ww = Sw + kPsdWsz;
  const int nblk = H * (H + 1) / 2;
  unsigned bar_target = 0;
  auto gbar = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are acknowledged
    __syncthreads();
    if (G > 1) {
      bar_target += (unsigned)G;
      if (tid == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bar_target) {
          __builtin_amdgcn_s_sleep(1);
          // a member never arrived (spin_budget: seconds by default): raise the error flag — the host restarts the solve with one
          // workgroup per matrix (work_residuals.inl spin_fallback) — and stop waiting, here and at every later barrier of every group of
          // this launch: what is computed from now on is thrown away, it only has to end soon
          if (++spins > spin_budget) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          if ((spins & 63) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
      }
      __syncthreads();
    }
  };

  bool wt = false;  // write-through (sc1) stores: set behind the first stopping test (below)
  auto st_shared = [&](double *p, double v) {
    if (wt) st_agent(p, v);
    else *p = v;
  };
#if PSD_PROFILE
  double prof[8] = {0., 0., 0., 0., 0., 0., 0., 0.};  // [1] norms [2] pivots from memory [3] barrier [4] tasks (+ pivots ahead) [5] barrier [6] feeder tasks [7] solve
#endif
  const bool la = look_ahead && G > 1 && H <= 8 * G && H >= 2;
  double *Acur = A, *Anxt = la ? A + 2 * (size_t)NP * NP : A;  // without look-ahead the update is in place
  // look-ahead roles: pivot k' of the next step on member k' % G, wave k' / G (< 8); the other wavefronts are task workers
  const int npw = g < H ? (H - g + G - 1) / G : 0;
  const bool pivot_wave = wave < npw;
  int wid = wave - npw;
  for (int g2 = 0; g2 < g; ++g2) wid += kPsdWaves - (g2 < H ? (H - g2 + G - 1) / G : 0);
  const int nworkers = G * kPsdWaves - H;

  // operands of task (k, k2 >= k) of the current step: the block rows{p,q} x cols{p2,q2} of Acur and the two logged rotations
  auto task_pairs = [&](int task, int &k, int &k2) {
    int kk = 0, rem = task;
    while (rem >= H - kk) { rem -= H - kk; ++kk; }
    k = kk;
    k2 = kk + rem;
  };
  auto load_task = [&](int k, int k2, int step, double (&av)[4], double (&w1)[4], double (&w2)[4]) {
    const int p = osch[2 * k], q = osch[2 * k + 1], p2 = osch[2 * k2], q2 = osch[2 * k2 + 1];
    const double *W1 = Wlog + ((size_t)step * H + k) * kPsdWsz, *W2 = Wlog + ((size_t)step * H + k2) * kPsdWsz;
    const int row = pq_index(li, p, q);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      av[kk] = ld_agent(&Acur[row + (size_t)ld * pq_index(4 * kk + lk, p2, q2)]);
      w2[kk] = ld_agent(&W2[psd_log_at(4 * kk + lk, li)]);
      w1[kk] = ld_agent(&W1[psd_log_at(4 * kk + lk, li)]);
    }
  };
  auto mma_task = [&](const double (&av)[4], const double (&w1)[4], const double (&w2)[4]) {
    f64x4 T = {0., 0., 0., 0.};  // T = Bm * W2
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) T = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], w2[kk], T, 0, 0, 0);
    f64x4 Rr = {0., 0., 0., 0.};  // R = W1' * T: lane holds R[lk + 4t][li], rows <-> pair k, columns <-> pair k2
#pragma unroll
    for (int t = 0; t < 4; ++t) Rr = __builtin_amdgcn_mfma_f64_16x16x4f64(w1[t], T[t], Rr, 0, 0, 0);
    return Rr;
  };
  // Two tasks per trip: the loads of both are in flight together.
  auto run_tasks = [&](int first, int stride, int step) {
    for (int task0 = first; task0 < nblk; task0 += 2 * stride) {
      int tk[2], tk2[2];
      bool live[2];
      double av[2][4], w1[2][4], w2[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int task = task0 + u * stride;
        live[u] = task < nblk;
        task_pairs(live[u] ? task : 0, tk[u], tk2[u]);
        if (live[u]) load_task(tk[u], tk2[u], step, av[u], w1[u], w2[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!live[u]) break;
        const f64x4 Rr = mma_task(av[u], w1[u], w2[u]);
        const int p = osch[2 * tk[u]], q = osch[2 * tk[u] + 1], p2 = osch[2 * tk2[u]], q2 = osch[2 * tk2[u] + 1];
        const int gi = pq_index(li, p2, q2), row = pq_index(li, p, q);
        if (tk[u] != tk2[u]) {
#pragma unroll
          for (int t = 0; t < 4; ++t) st_shared(&Anxt[gi + (size_t)ld * pq_index(lk + 4 * t, p, q)], Rr[t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = Rr[t];
        wave_sync();
#pragma unroll
        for (int t = 0; t < 4; ++t) st_shared(&Anxt[row + (size_t)ld * pq_index(lk + 4 * t, p2, q2)], Sw[li + 17 * (lk + 4 * t)]);
        wave_sync();
      }
    }
  };
  auto log_w = [&](int step, int k) {
    double *Wk = Wlog + ((size_t)step * H + k) * kPsdWsz;
    for (int e = lane; e < 256; e += 64) st_shared(&Wk[e], Ww[(e >> 4) + kPsdWLd * (e & 15)]);
  };
  // pivot kn of the NEXT step from the three tasks of this step that produce its block
  auto pivot_ahead = [&](int kn, int step) {
    const int pn = osch_n[2 * kn], qn = osch_n[2 * kn + 1];
    const int a = where[pn] >> 1, pa = where[pn] & 1, b = where[qn] >> 1, pb = where[qn] & 1;
    PSD_TICK(t_f0);
    // the three tasks share two rotations: W_a, W_b (20 loads in flight, not 36)
    double av[3][4], wa[4], wb[4];
    {
      const int pa_ = osch[2 * a], qa_ = osch[2 * a + 1], pb_ = osch[2 * b], qb_ = osch[2 * b + 1];
      const int pl = a < b ? pa_ : pb_, ql = a < b ? qa_ : qb_, ph = a < b ? pb_ : pa_, qh = a < b ? qb_ : qa_;
      const double *Wa = Wlog + ((size_t)step * H + a) * kPsdWsz, *Wb = Wlog + ((size_t)step * H + b) * kPsdWsz;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        av[0][kk] = ld_agent(&Acur[pq_index(li, pa_, qa_) + (size_t)ld * pq_index(4 * kk + lk, pa_, qa_)]);
        av[1][kk] = ld_agent(&Acur[pq_index(li, pb_, qb_) + (size_t)ld * pq_index(4 * kk + lk, pb_, qb_)]);
        av[2][kk] = ld_agent(&Acur[pq_index(li, pl, ql) + (size_t)ld * pq_index(4 * kk + lk, ph, qh)]);
        wa[kk] = ld_agent(&Wa[psd_log_at(4 * kk + lk, li)]);
        wb[kk] = ld_agent(&Wb[psd_log_at(4 * kk + lk, li)]);
      }
    }
    const f64x4 Ra = mma_task(av[0], wa, wa), Rb = mma_task(av[1], wb, wb);
    const f64x4 Rc = a < b ? mma_task(av[2], wa, wb) : mma_task(av[2], wb, wa);
    // S = [[A(pn,pn), A(pn,qn)], [A(qn,pn), A(qn,qn)]] after this step; R[x][y] = A[pair k index x][pair k2 index y]
    // Branch-free: every lane writes all four candidates, the ones that do not belong to the pivot go to the lane's own
    // slot of the (not yet initialised) W scratch.  row = lk + 4t, col = li: row >> 3 = t >> 1.
    const int pr = a < b ? pa : pb, pc = a < b ? pb : pa;  // position of the cross block's row / column 8-block in Rc
    const int c7 = li & 7, ch = li >> 3, trash = kPsdWsz + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rh = t >> 1, r7 = lk + 4 * (t & 1);
      const int ia = (rh == pa && ch == pa) ? r7 + kPsdLd * c7 : trash;
      const int ib = (rh == pb && ch == pb) ? 8 + r7 + kPsdLd * (8 + c7) : trash;
      const bool inc = rh == pr && ch == pc;
      const int ic1 = !inc ? trash : (a < b ? r7 + kPsdLd * (8 + c7) : 8 + r7 + kPsdLd * c7);
      const int ic2 = !inc ? trash : (a < b ? 8 + c7 + kPsdLd * r7 : c7 + kPsdLd * (8 + r7));
      Sw[ia] = Ra[t];
      Sw[ib] = Rb[t];
      Sw[ic1] = Rc[t];
      Sw[ic2] = Rc[t];
    }
    wave_sync();
    PSD_TICK(t_f1);
    wave_jacobi16(Sw, Ww, lane, false);  // step + 1 is never the first step of a sweep
    PSD_TICK(t_f2);
    log_w(step + 1, kn);
    PSD_ACC(6, t_f0, t_f1);
    PSD_ACC(7, t_f1, t_f2);
  };

  int nlog = 0;
  for (int sweep = 0; sweep < kPsdMaxSweeps; ++sweep) {
    PSD_TICK(t_n0);
    // every member evaluates the stopping test on the same data in the same order: the same decision everywhere
    // (the first test comes before anything spins: a matrix with nothing to do — refinable as it arrives, the steady state of ADMM, or refined to
    //  the relaxed level — leaves without a barrier)
    const int code = psd_stop_test<true>(Acur, ld, n, lds, red, bc, offtol2, stop_mode, R, state, g == 0, Vt, round == 0 && sweep == 0);
    PSD_TICK(t_n1);
    PSD_ACC(1, t_n0, t_n1);
    if (code != 0) {
      if (g == 0) {
        if (tid == 0) {
          state[1] = 1.;
          if (code == 2) state[7] = 1.;
        }
        if (code == 2)
          for (int j = tid; j < NP; j += kPsdThreads) lam[j] = j < n ? lds[j] : 0.;  // the diagonal the gate saw (COMM's epilogue)
      }
      break;
    }
    if (sweep == 0) {
      // one XCD for the whole group?  (state[6]: OR of the members' XCD bits, zeroed by the front kernel)
      if (G > 1) {
        unsigned *xmask = reinterpret_cast<unsigned *>(state + 6);
        if (tid == 0) __hip_atomic_fetch_or(xmask, 1u << psd_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gbar();
        if (tid == 0) bc[1] = (double)__popc(__hip_atomic_load(xmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        wt = bc[1] != 1.;
#ifdef PSD_MC_FORCE_WT
        wt = true;
#endif
      }
      // (behind the group's first barrier: every member has read state[] by now — a member that starts late must not see this)
      if (refined && g == 0 && tid == 0) {  // the refinement left too much: back to the sweeps
        state[1] = 0.;
        state[7] = 2.;
        state[10] += 1.;
      }
    }
    if (sweep >= kPsdLogSweeps) break;  // log full: the next round continues

    for (int r = 0; r < NB - 1; ++r) {
      const bool ahead = la && r + 1 < NB - 1;
      for (int k = tid; k < H; k += kPsdThreads) {
        int p, q;
        rr_pair(r, k, NB, p, q);
        osch[2 * k] = p;
        osch[2 * k + 1] = q;
        if (la) { where[p] = 2 * k; where[q] = 2 * k + 1; }
        if (ahead) {
          rr_pair(r + 1, k, NB, p, q);
          osch_n[2 * k] = p;
          osch_n[2 * k + 1] = q;
        }
      }
      __syncthreads();
      PSD_TICK(t_p0);
      if (!la || r == 0) {
        // ---------------- pivots of this step from memory: pivot k on workgroup k % G, wave k / G ----------------
        for (int k = g + G * wave; k < H; k += G * kPsdWaves) {
          const int p = osch[2 * k], q = osch[2 * k + 1];
          for (int e = lane; e < 256; e += 64) {
            const int i = e & 15, j = e >> 4;
            Sw[i + kPsdLd * j] = ld_agent(&Acur[pq_index(i, p, q) + (size_t)ld * pq_index(j, p, q)]);
          }
          wave_sync();
          wave_jacobi16(Sw, Ww, lane, r == 0);
          log_w(nlog, k);
        }
#if PSD_PROFILE
        __syncthreads();
#endif
        PSD_TICK(t_p1);
        gbar();
        PSD_TICK(t_p2);
        PSD_ACC(2, t_p0, t_p1);
        PSD_ACC(3, t_p1, t_p2);
      }
      PSD_TICK(t_p3);
      // ---------------- A_{t+1} = W' A_t W over block pairs k <= k2 (and, ahead, the next step's pivots) ----------------
      if (ahead && pivot_wave) {
        // the pivot chain is the critical path of the step: its instructions go first on the SIMD it shares with three workers
        __builtin_amdgcn_s_setprio(3);
        pivot_ahead(g + G * wave, nlog);
        __builtin_amdgcn_s_setprio(0);
      } else {
        run_tasks(ahead ? wid : g + G * wave, ahead ? nworkers : G * kPsdWaves, nlog);
      }
#if PSD_PROFILE
      __syncthreads();
#endif
      PSD_TICK(t_p4);
      gbar();
      PSD_TICK(t_p5);
      PSD_ACC(4, t_p3, t_p4);
      PSD_ACC(5, t_p4, t_p5);
      if (la) { double *t = Acur; Acur = Anxt; Anxt = t; }
      ++nlog;
    }
  }
  if (Acur != A) {  // an odd number of steps: the matrix goes home (the later kernels read A)
    for (size_t e = (size_t)g * kPsdThreads + tid; e < (size_t)NP * NP; e += (size_t)G * kPsdThreads) A[e] = ld_agent(&Acur[e]);
  }
  if (g == 0 && tid == 0) state[2] = (double)nlog;
#if PSD_PROFILE
  if (cidx == 0 && tid == 0 && nlog > 0)
    printf("  mc member %d round %d: %d steps; per step (us): norms %.2f  pivots(mem) %.2f  barrier %.2f  tasks %.2f  barrier %.2f\n", g, round,
           nlog, prof[1] / 100 / nlog, prof[2] / 100 / nlog, prof[3] / 100 / nlog, prof[4] / 100 / nlog, prof[5] / 100 / nlog),
    printf("      pivot ahead (wave 0): feeder tasks %.2f  solve %.2f us per step\n", prof[6] / 100 / nlog, prof[7] / 100 / nlog);
#endif
}

// Split mode, V <- V W_1 W_2 ... : one 512-lane workgroup per (matrix, 16-row strip of V).  The strip lives in LDS
// (16 x NP doubles); step t rotates, for every pivot (p,q) of that step, the strip's 16 columns {block p, block q}
// by the logged 16x16 W — 4 MFMAs per pivot, the same instruction sequence as the in-kernel V tasks of MODE 0.
#ifndef PSD_APPLY_THREADS
#define PSD_APPLY_THREADS 512  // 8 wavefronts share the 13 pivots of a step (tools/dbg/psd_apply_threads.sh: 257 us per round at 256 lanes, 212 at 512, 217 at 1024)
#endif
constexpr int kPsdApplyThreads = PSD_APPLY_THREADS;
__global__ __launch_bounds__(kPsdApplyThreads) void k_psd_apply_v(PsdBatch B, double *scratch, const int *stall) {
  SCS_STALL_GUARD(stall);
  const PsdWg wg = psd_wg(B.count);
  if (!wg.ok) return;
  extern __shared__ __attribute__((aligned(16))) double strip[];  // [row + 16 * col]
  const int n = B.order[wg.cidx];
  if (n < 2) return;
  const int NP = (int)psd_np(n), NB = NP / kPsdB, H = NB / 2, ld = NP, ntile = NP / 16;
  const int rt = wg.bx;
  if (rt >= ntile) return;
  double *A = scratch + B.woff[wg.cidx];
  double *V = A + (size_t)NP * NP;
  const double *Wlog = V + 3 * (size_t)NP * NP + (size_t)H * kPsdWsz;
  const double *state = Wlog + psd_log_doubles(n) + NP;
  const int nsteps = (int)state[2];
  if (nsteps == 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  for (int e = tid; e < 16 * NP; e += kPsdApplyThreads) strip[e] = V[(rt * 16 + (e & 15)) + (size_t)ld * (e >> 4)];
  __syncthreads();
  for (int t = 0; t < nsteps; ++t) {
    const int r = t % (NB - 1);
    for (int k = wave; k < H; k += kPsdApplyThreads / 64) {
      int p, q;
      rr_pair(r, k, NB, p, q);
      const double *W1 = Wlog + ((size_t)t * H + k) * kPsdWsz;
      double av[4], w[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        av[kk] = strip[li + 16 * pq_index(4 * kk + lk, p, q)];
        w[kk] = W1[psd_log_at(4 * kk + lk, li)];
      }
      f64x4 T = {0., 0., 0., 0.};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) T = __builtin_amdgcn_mfma_f64_16x16x4f64(w[kk], av[kk], T, 0, 0, 0);  // (V_blk W)'
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) strip[li + 16 * pq_index(lk + 4 * tt, p, q)] = T[tt];
    }
    __syncthreads();  // the next step pairs the column blocks differently
  }
  for (int e = tid; e < 16 * NP; e += kPsdApplyThreads) V[(rt * 16 + (e & 15)) + (size_t)ld * (e >> 4)] = strip[e];
}

// Split mode: F = Pi_+(D + E) of the nearly diagonal A (psd_fmap, see the reconstruction) once, element by element, into the
// buffer of V' (free after the warm-start GEMMs) — the B operand of the R1 GEMM.
__global__ __launch_bounds__(256) void k_psd_fmap(PsdBatch B, double *scratch, const int *stall) {
  SCS_STALL_GUARD(stall);
  const PsdWg wg = psd_wg(B.count);
  if (!wg.ok) return;
  const int n = B.order[wg.cidx];
  if (n < 2) return;
  const int NP = (int)psd_np(n), ld = NP;
  const double *A = scratch + B.woff[wg.cidx];
  double *F = scratch + B.woff[wg.cidx] + 3 * (size_t)NP * NP;
  for (long e = (long)wg.bx * 256 + threadIdx.x; e < (long)NP * NP; e += (long)wg.per * 256) {
    const int r = (int)(e % NP), k = (int)(e / NP);
    const double dj = r < n ? A[r + (size_t)ld * r] : 0., dk = k < n ? A[k + (size_t)ld * k] : 0.;
    F[e] = psd_fmap(A[e], dj, dk, r == k);
  }
}

// Refinement stage, V <- V Q in place: one 512-lane workgroup per (matrix, 16-row strip of V), the strip in LDS (as k_psd_apply_v);
// out[i][j] = sum_k strip[i][k] Q'[j][k], Q' (Tm) read down its columns; the new strip is assembled in a second LDS buffer.
__global__ __launch_bounds__(kPsdApplyThreads) void k_psd_apply_q(PsdBatch B, double *scratch, const int *stall) {
  SCS_STALL_GUARD(stall);
  const PsdWg wg = psd_wg(B.count);
  if (!wg.ok) return;
  extern __shared__ __attribute__((aligned(16))) double strip[];  // [row + 16 * col], twice
  const int n = B.order[wg.cidx];
  if (n < 2) return;
  const int NP = (int)psd_np(n), H = NP / kPsdB / 2, ld = NP, ntile = NP / 16;
  const int rt = wg.bx;
  if (rt >= ntile) return;
  double *A = scratch + B.woff[wg.cidx];
  double *V = A + (size_t)NP * NP;
  const double *Qt = V + 2 * (size_t)NP * NP;  // (Vt)
  double *state = A + 4 * (size_t)NP * NP + (size_t)H * kPsdWsz + psd_log_doubles(n) + NP;
  if (state[7] != 1.) return;
  double *out = strip + 16 * (size_t)NP;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  for (int e = tid; e < 16 * NP; e += kPsdApplyThreads) strip[e] = V[(rt * 16 + (e & 15)) + (size_t)ld * (e >> 4)];
  __syncthreads();
  // a wavefront takes a PAIR of column tiles: one 16-byte load per lane brings the Q' operands of both (columns 2 li, 2 li + 1 of the
  // pair; k_psd_gemm has the reasoning), so the 13 column tiles of an order-200 matrix are one round of the 8 wavefronts
  const int ldh = ld / 2;
  for (int tp = wave; 2 * tp < ntile; tp += kPsdApplyThreads / 64) {
    const f64x2 *pb = reinterpret_cast<const f64x2 *>(Qt + tp * 32 + 2 * li + (size_t)ld * lk);
    f64x4 acc[2] = {{0., 0., 0., 0.}, {0., 0., 0., 0.}};
    f64x2 bq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bq[u] = pb[(size_t)ldh * (4 * u)];
    for (int k0 = 0; k0 < NP; k0 += 16) {  // four k-steps per trip, the next trip's operands in flight (the last trip re-fetches)
      f64x2 bn[4];
      const int kn = min(k0 + 16, NP - 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) bn[u] = pb[(size_t)ldh * (kn + 4 * u)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double a = strip[li + 16 * (k0 + 4 * u + lk)];
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[u].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[u].y, acc[1], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) bq[u] = bn[u];
    }
    // lane holds out[row = lk + 4t][col = tp * 32 + 2 li + cq]
#pragma unroll
    for (int cq = 0; cq < 2; ++cq) {
      const int col = tp * 32 + 2 * li + cq;
      if (col < NP) {
#pragma unroll
        for (int t = 0; t < 4; ++t) out[(lk + 4 * t) + 16 * col] = acc[cq][t];
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < 16 * NP; e += kPsdApplyThreads) V[(rt * 16 + (e & 15)) + (size_t)ld * (e >> 4)] = out[e];
  if (rt == 0 && tid == 0) state[9] += 1.;  // calls refined
}

// Split mode: the four GEMM phases as multi-workgroup launches — grid (workgroups per matrix, matrices), 4 wavefronts per
// workgroup.  A wavefront's task is a block of kPsdRT row tiles x kPsdNJ column tiles: the operand fetches, not the matrix
// cores, bound these launches (every k-step of one row tile x 4 column tiles issues 5 loads for 4 MFMAs and the texture
// addresser takes ~35 clocks per load instruction against 16 clocks per MFMA and CU), so two row tiles share the four
// B operands — 6 loads for 8 MFMAs — and the k-steps are software-pipelined (four in flight).  Measured at 50 x order 200
// (tools/dbg/psd_gemm_threads.sh): 44 us per launch with 1 x 4 tiles and no pipeline, 53 us with 2 x 4 tiles alone (fewer
// wavefronts, every k-step one L2 round trip), 38 us with both; the workgroup size (64 / 128 / 256 lanes) does not matter.
// Every output tile still accumulates its k-steps in order: the bits of psd_task_*.
// Round 5, the refinement stage (see psd_stop_test) as four more kinds, all gated by state[7] == 1:
//   COMM  K2 = mixed((2 S - (S K1' + K1 S')) ./ den)  — the two products accumulate into one tile; S carries its diagonal, which
//         contributes (d_j - d_i) K1_ij = S_ij to the sum: [S_off, K1]_ij = S_ij - acc_ij on the mixed pairs.  Tiles on and below the
//         diagonal, mirrored with the opposite sign (diagonal tiles antisymmetrised): K2 is EXACTLY antisymmetric.      A, Vt -> Tm
//         (K1: written into Vt by the sweep kernel's gate, psd_stop_test)
//   KK    Q' = I - K2 + K2^2 / 2 = I - K2 - (K2 K2') / 2, lower tiles + mirror: the symmetric part exactly symmetric    Tm -> Vt
//   T     T' = (S Q)' stored transposed (the B-operand layout of the next product)                                      A, Vt -> Tm
//   S1    S1 = Q' T, lower tiles + mirror + symmetrised diagonal tiles (the epilogue of G2)                              Vt, Tm -> A
//   (Tm is the sweeps' second copy of A while they run: it is free again when the stage starts, and the round behind it finds Q' dead)
enum : int { PSD_G1 = 0, PSD_G2, PSD_R1, PSD_R2, PSD_COMM, PSD_KK, PSD_T, PSD_S1 };
// Round 5: who computes what.  A launch was 28 tasks of 2 x 4 tiles per order-200 matrix, two wavefronts per workgroup: 1400 wavefronts
// on 1024 SIMDs, so the launch took as long as the SIMDs that got two of them — 2 x 8 tiles x 52 k-steps x 64 clocks = 22 us of matrix-core
// time, whatever the operand fetches did (16-byte loads, a real software pipeline: no change) — and the kinds that only need the lower
// triangle took as long as the full ones.  Now a task is a ROW PAIR x a column group of about five tiles (the trailing single row of an
// odd tile count: groups of about eight), ONE wavefront = one workgroup per task, so that the tasks of a launch fit the SIMDs one each:
// order 200 (13 tiles): 6 x 3 + 2 = 20 tasks of <= 10 tiles per matrix, 1000 per launch of 50; lower-triangular kinds 14.
// Tasks are dealt to the XCDs in contiguous runs (psd_gemm_grid): an L2 sees the operands of ~1/8 of the matrices.
constexpr int kPsdGemmThreads = 64;
constexpr int kPsdRT = 2;
constexpr int kPsdPf = 4;      // k-steps per trip of the software pipeline
constexpr int kPsdNJ2max = 4;  // column tiles of a task come in pairs (one 16-byte load): at most 8
__host__ __device__ inline int psd_gemm_ncg(int ntile) { return (ntile + 4) / 5; }           // column groups of a row pair
__host__ __device__ inline int psd_gemm_ncg1(int ntile) { return (ntile + 7) / 8; }          // ... of the trailing single row
__host__ __device__ inline int psd_gemm_cstart(int g, int ncg, int ntile) { return g >= ncg ? ntile : 2 * ((g * ntile) / (2 * ncg)); }
__host__ __device__ inline int psd_gemm_cstart1(int g, int ncg, int ntile) { return g >= ncg ? ntile : 2 * ((2 * g * ntile + 2 * ncg) / (4 * ncg)); }  // (rounded: widths <= 8)
__host__ __device__ inline int psd_gemm_tasks(int ntile) { return (ntile / 2) * psd_gemm_ncg(ntile) + (ntile & 1) * psd_gemm_ncg1(ntile); }
// 1-D grid of 8 * chunk workgroups, chunk = ceil(count * per / 8): workgroup id -> XCD (id & 7) -> global task xcd * chunk + (id >> 3)
__host__ inline unsigned psd_gemm_grid(int per, int count) { return 8u * (unsigned)(((long)count * per + 7) / 8); }

template <int KIND>
__global__ __launch_bounds__(kPsdGemmThreads) void k_psd_gemm(double *x, PsdBatch B, double *scratch, int allow_warm, const int *stall, int per) {
  SCS_STALL_GUARD(stall);
  __shared__ double Sw[16 * 17];
  __shared__ double Rw[32 * 33];  // a tile pair x column pair of accumulators on their way back to whole tiles
  const int chunk = (int)gridDim.x >> 3;
  const int gtask = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  const int cidx = gtask / per, task = gtask - cidx * per;
  if (cidx >= B.count) return;
  const int n = B.order[cidx];
  if (n < 2) return;  // orders 0 and 1 were finished by the front kernel
  const int NP = (int)psd_np(n), ntile = NP / 16, H = NP / kPsdB / 2, ld = NP;
  if (task >= psd_gemm_tasks(ntile)) return;  // (`per` is the task count of the largest matrix of the batch)
  double *A = scratch + B.woff[cidx];
  double *V = A + (size_t)NP * NP;
  double *Tm = V + (size_t)NP * NP;
  double *Vt = Tm + (size_t)NP * NP;  // V' for G1 / G2; F = Pi_+(D + E) (k_psd_fmap) for R1
  double *state = Vt + (size_t)NP * NP + (size_t)H * kPsdWsz + psd_log_doubles(n) + NP;
  const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
  const bool warm = allow_warm && state[0] >= 1.;
  if ((KIND == PSD_G1 || KIND == PSD_G2) && !warm) return;  // cold start: A0 = A
  if (KIND >= PSD_COMM && state[7] != 1.) return;            // no refinement stage for this matrix in this call
  constexpr bool lower = KIND == PSD_G2 || KIND == PSD_R2 || KIND == PSD_COMM || KIND == PSD_KK || KIND == PSD_S1;  // (anti)symmetric results: tiles on and below the diagonal only
  constexpr int npass = KIND == PSD_COMM ? 2 : 1;
  const double *lam = state - NP;
  // C[i][j] = sum_k Aop[i][k] Bop[j][k]:  G1 Tt = Vt A (A symmetric)   G2 A0 = Vt Tt'   R1 T = V F   R2 X+ = T V'
  //   COMM S K1' + K1 S' (two passes)   KK K2 K2'   T S Q   S1 Q' T
  const double *Aop = KIND == PSD_G1 || KIND == PSD_G2 || KIND == PSD_S1 ? Vt : KIND == PSD_R1 ? V : KIND == PSD_COMM || KIND == PSD_T ? A : Tm;
  const double *Bop = KIND == PSD_G1 ? A : KIND == PSD_G2 || KIND == PSD_KK || KIND == PSD_S1 ? Tm : KIND == PSD_R1 || KIND == PSD_COMM || KIND == PSD_T ? Vt : V;
  double *Ko = KIND == PSD_COMM || KIND == PSD_T ? Tm : Vt;  // where COMM / T (Tm) and KK (Vt) leave their result; KK reads K2 from Tm
  double *X = x + B.off[cidx];
  const double sq2 = 1.41421356237309504880;
  // task -> (row pair | trailing single row, column group)
  const int npair = ntile / 2, ncg = psd_gemm_ncg(ntile), ncg1 = psd_gemm_ncg1(ntile);
  int ti0, rows, tj0, tj1;
  if (task < npair * ncg) {
    const int rp = task / ncg, g = task - rp * ncg;
    ti0 = 2 * rp;
    rows = 2;
    tj0 = psd_gemm_cstart(g, ncg, ntile);
    tj1 = psd_gemm_cstart(g + 1, ncg, ntile);
  } else {
    const int g = task - npair * ncg;
    ti0 = ntile - 1;
    rows = 1;
    tj0 = psd_gemm_cstart1(g, ncg1, ntile);
    tj1 = psd_gemm_cstart1(g + 1, ncg1, ntile);
  }
  const int tilast = ti0 + rows - 1;
  if (lower && tj0 > tilast) return;
  if (lower) tj1 = min(tj1, tilast + 1);
  const int nj2 = (tj1 - tj0 + 1) / 2;  // column-tile pairs of this task (1 .. kPsdNJ2max)
  f64x4 acc[kPsdRT][2 * kPsdNJ2max];
#pragma unroll
  for (int j = 0; j < 2 * kPsdNJ2max; ++j)
#pragma unroll
    for (int r = 0; r < kPsdRT; ++r) acc[r][j] = f64x4{0., 0., 0., 0.};
  // Operand fetches (round 5): ONE 16-byte load per lane brings the A operands of both row tiles and one brings the B operands of
  // two column tiles — lane li holds rows 2 li and 2 li + 1 of the 32 rows of the tile pair, so the "tiles" the matrix cores work on
  // are the even and the odd rows (columns) of the pair; which row a lane feeds does not enter an output element's arithmetic
  // (its four k-terms are added in the same order), so every element keeps its bits.  The accumulators come out interleaved the
  // same way and are put back into whole tiles through LDS before the epilogue.  Rows / columns beyond the matrix (the odd tile
  // at the edge) are read from whatever follows in the scratch — finite numbers that only reach accumulator entries the
  // epilogue skips.
  const int ldh = ld / 2;  // (NP is a multiple of 16; every operand buffer starts on a 16-byte boundary)
  // Software pipeline over the k-steps, a TRIP = kPsdPf k-steps: two register sets, the loads of the next trip are issued before the
  // MFMAs of the current one.  Every load is unconditional (the tail re-fetches the last trip: a conditional load in this loop makes
  // hipcc fall back to s_waitcnt vmcnt(0)).  (One set carried around the back edge and refilled behind its use was rotated by hipcc
  // into load - wait - use: s_waitcnt vmcnt(9) right behind the twelve loads of a trip.)
  auto product = [&](auto rows_tag, auto nj2_tag) {
    constexpr int ROWS = decltype(rows_tag)::value, NJ2 = decltype(nj2_tag)::value;
#pragma unroll
    for (int pass = 0; pass < npass; ++pass) {
      const double *Ao = pass == 0 ? Aop : Bop, *Bo = pass == 0 ? Bop : Aop;  // (COMM: the second product has the operands swapped)
      const f64x2 *pa2 = reinterpret_cast<const f64x2 *>(Ao + ti0 * 16 + 2 * li + (size_t)ld * lk);
      const double *pa1 = Ao + ti0 * 16 + li + (size_t)ld * lk;
      const f64x2 *pb2[NJ2];
#pragma unroll
      for (int j = 0; j < NJ2; ++j) pb2[j] = reinterpret_cast<const f64x2 *>(Bo + tj0 * 16 + 32 * j + 2 * li + (size_t)ld * lk);
      f64x2 a0[kPsdPf], b0[kPsdPf][NJ2], a1[kPsdPf], b1[kPsdPf][NJ2];
      auto load = [&](f64x2 (&as)[kPsdPf], f64x2 (&bs)[kPsdPf][NJ2], int trip) {
#if defined(PSD_GEMM_ABL) && PSD_GEMM_ABL == 2  // (lab) MFMAs only: the operands of the first trip, fetched once
        if (trip > 0) return;
#endif
#pragma unroll
        for (int st = 0; st < kPsdPf; ++st) {
          const size_t at = (size_t)ldh * (4 * (kPsdPf * trip + st));
          if (ROWS == 2) as[st] = pa2[at];
          else as[st].x = pa1[2 * at];  // the trailing single row tile: lane li feeds row li, as in the one-workgroup kernels
#pragma unroll
          for (int j = 0; j < NJ2; ++j) bs[st][j] = pb2[j][at];
        }
      };
      auto use = [&](const f64x2 (&as)[kPsdPf], const f64x2 (&bs)[kPsdPf][NJ2]) {
#if defined(PSD_GEMM_ABL) && PSD_GEMM_ABL == 1  // (lab, tools/psd_lab.hip: profiles/r05_psd_gemm_ablation.txt) loads only: one cheap use per loaded value
#pragma unroll
        for (int st = 0; st < kPsdPf; ++st) {
          acc[0][0][0] += as[st].x + as[st].y;
#pragma unroll
          for (int j = 0; j < NJ2; ++j) acc[0][0][1] += bs[st][j].x + bs[st][j].y;
        }
        return;
#endif
#pragma unroll
        for (int st = 0; st < kPsdPf; ++st) {
          const double a[kPsdRT] = {as[st].x, as[st].y};
#pragma unroll
          for (int j = 0; j < NJ2; ++j)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
              acc[r][2 * j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], bs[st][j].x, acc[r][2 * j], 0, 0, 0);
              acc[r][2 * j + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], bs[st][j].y, acc[r][2 * j + 1], 0, 0, 0);
            }
        }
      };
      const int ntrips = NP / (4 * kPsdPf);  // NP is a multiple of 16 = 4 kPsdPf
      load(a0, b0, 0);
      int trip = 0;
      for (; trip + 1 < ntrips; trip += 2) {
        load(a1, b1, trip + 1);
        use(a0, b0);
        load(a0, b0, min(trip + 2, ntrips - 1));
        use(a1, b1);
      }
      if (trip < ntrips) use(a0, b0);  // an odd number of trips: the last one sits in the first set
    }
  };
  using std::integral_constant;
  if (rows == 2) {
    if (nj2 == 1) product(integral_constant<int, 2>{}, integral_constant<int, 1>{});
    else if (nj2 == 2) product(integral_constant<int, 2>{}, integral_constant<int, 2>{});
    else product(integral_constant<int, 2>{}, integral_constant<int, 3>{});
  } else {
    if (nj2 <= 2) product(integral_constant<int, 1>{}, integral_constant<int, 2>{});
    else product(integral_constant<int, 1>{}, integral_constant<int, 4>{});
  }
  // acc[r][j][t] = C[row 2 (lk + 4t) + r of the tile pair (single row tile: row lk + 4t)][column 2 li + (j & 1) of column pair j >> 1]
#if defined(PSD_GEMM_ABL) && PSD_GEMM_ABL == 3  // (lab) no epilogue: a store that never happens keeps the accumulators alive
  {
    double t = 0.;
#pragma unroll
    for (int j = 0; j < 2 * kPsdNJ2max; ++j)
#pragma unroll
      for (int r = 0; r < kPsdRT; ++r) t += acc[r][j][0] + acc[r][j][1] + acc[r][j][2] + acc[r][j][3];
    if (t == 1.2345e-300) Tm[lane] = t;
    return;
  }
#endif
#pragma unroll
  for (int jp = 0; jp < kPsdNJ2max; ++jp) {
    if (tj0 + 2 * jp >= tj1) break;  // (uniform: nothing of this column pair is wanted)
    // COMM / KK: what the epilogues of this column pair's (up to four) tiles read from memory — S and the eigenvalues, K2 at the entry
    // and at its mirror — is requested HERE, in front of the LDS re-layout, instead of tile by tile in front of its use (round 5, late:
    // these two kinds ran 10 us longer than the others with the same products; the values are the same, so are the bits)
    double pf_a[kPsdRT][2][4], pf_b[kPsdRT][2][4], pf_l[2];
    if constexpr (KIND == PSD_KK || KIND == PSD_COMM) {
#pragma unroll
      for (int r = 0; r < kPsdRT; ++r) {
        const int ti = ti0 + r;
        if (r >= rows) break;
#pragma unroll
        for (int cq = 0; cq < 2; ++cq) {
          const int tj = tj0 + 2 * jp + cq;
          if (tj >= tj1 || (lower && tj > ti)) break;
          if constexpr (KIND == PSD_COMM) pf_l[cq] = lam[tj * 16 + li];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if constexpr (KIND == PSD_KK) {
              pf_a[r][cq][t] = ti != tj ? Tm[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] : 0.;
              pf_b[r][cq][t] = Tm[(ti * 16 + li) + (size_t)ld * (tj * 16 + lk + 4 * t)];
            } else {
              const int i = ti * 16 + lk + 4 * t;
              pf_a[r][cq][t] = A[i + (size_t)ld * (tj * 16 + li)];
              pf_b[r][cq][t] = lam[i];
            }
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kPsdRT; ++r)
#pragma unroll
      for (int cq = 0; cq < 2; ++cq)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = rows == 2 ? 2 * (lk + 4 * t) + r : 16 * r + lk + 4 * t;
          Rw[row + 33 * (2 * li + cq)] = acc[r][2 * jp + cq][t];
        }
    wave_sync();
#pragma unroll
    for (int r = 0; r < kPsdRT; ++r) {
      const int ti = ti0 + r;
      if (r >= rows) break;
#pragma unroll
      for (int cq = 0; cq < 2; ++cq) {
        const int tj = tj0 + 2 * jp + cq;
        if (tj >= tj1 || (lower && tj > ti)) break;
        f64x4 c;  // lane holds C[row = lk + 4t][col = li] of tile (ti, tj)
#pragma unroll
        for (int t = 0; t < 4; ++t) c[t] = Rw[(16 * r + lk + 4 * t) + 33 * (16 * cq + li)];
          if (KIND == PSD_G1 || KIND == PSD_R1) {  // stored through the 16x17 transpose: li runs down the columns of the result
#pragma unroll
            for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = c[t];
            wave_sync();
#pragma unroll
            for (int t = 0; t < 4; ++t) Tm[(ti * 16 + li) + (size_t)ld * (tj * 16 + lk + 4 * t)] = Sw[li + 17 * (lk + 4 * t)];
            wave_sync();
          } else if (KIND == PSD_T) {  // T' straight from the C layout: li runs down a column of the transpose
#pragma unroll
            for (int t = 0; t < 4; ++t) Ko[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] = c[t];
          } else if (KIND == PSD_COMM) {
            double val[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const double di = pf_b[r][cq][t], dj = pf_l[cq];
              const double sij = pf_a[r][cq][t];
              val[t] = ((di > 0.) != (dj > 0.)) ? (2. * sij - c[t]) / (dj - di) : 0.;
            }
            if (ti != tj) {
#pragma unroll
              for (int t = 0; t < 4; ++t) Ko[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] = -val[t];  // K2_ji = -K2_ij
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = val[t];
            wave_sync();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int rr = li, cc = lk + 4 * t;
              double v = Sw[rr + 17 * cc];
              if (ti == tj) v = 0.5 * (v - Sw[cc + 17 * rr]);
              Ko[(ti * 16 + rr) + (size_t)ld * (tj * 16 + cc)] = v;
            }
            wave_sync();
          } else if (KIND == PSD_KK) {  // Q'_ij = delta_ij - K2_ij - acc_ij / 2 (acc = K2 K2' = -K2^2), both triangles from the lower tiles
            if (ti != tj) {
#pragma unroll
              for (int t = 0; t < 4; ++t) {  // the mirror entry (jc, i): its own K2 entry is read where it lies (coalesced)
                const size_t at = (tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t);
                Ko[at] = -pf_a[r][cq][t] - 0.5 * c[t];
              }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = c[t];
            wave_sync();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int rr = li, cc = lk + 4 * t;
              double v = Sw[rr + 17 * cc];
              if (ti == tj) v = 0.5 * (v + Sw[cc + 17 * rr]);
              const size_t at = (ti * 16 + rr) + (size_t)ld * (tj * 16 + cc);
              Ko[at] = ((ti == tj && rr == cc) ? 1. : 0.) - pf_b[r][cq][t] - 0.5 * v;
            }
            wave_sync();
          } else if (KIND == PSD_G2 || KIND == PSD_S1) {  // mirrored; diagonal tiles symmetrised (average of the two triangles): A0 exactly symmetric
            if (ti != tj) {
#pragma unroll
              for (int t = 0; t < 4; ++t) A[(tj * 16 + li) + (size_t)ld * (ti * 16 + lk + 4 * t)] = c[t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) Sw[(lk + 4 * t) + 17 * li] = c[t];
            wave_sync();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int rr = li, cc = lk + 4 * t;
              double v = Sw[rr + 17 * cc];
              if (ti == tj && rr != cc) v = 0.5 * ((rr > cc ? v : Sw[cc + 17 * rr]) + (rr > cc ? Sw[cc + 17 * rr] : v));
              A[(ti * 16 + rr) + (size_t)ld * (tj * 16 + cc)] = v;
            }
            wave_sync();
          } else {  // R2: lower-triangular tiles straight into the packed vector
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int i = ti * 16 + lk + 4 * t, jc = tj * 16 + li;
              if (i < n && jc <= i) {
                const long base = (long)jc * n - (long)jc * (jc - 1) / 2;
                X[base + (i - jc)] = (i == jc) ? c[t] : c[t] * sq2;
              }
            }
          }

      }
    }
    wave_sync();
  }
  if (KIND == PSD_R2 && task == 0 && lane == 0) state[0] = warm ? state[0] + 1. : 1.;
}

// ---------------------------------------------------------------------------
// Small matrices (order <= kPsdSmallMax = 32): ONE wavefront per matrix, everything in LDS.
// At this size the block algorithm above is all latency (3 outer steps x [16x16 pivot solve + update + 3
// workgroup barriers] per sweep, ~30 us); a plain parallel-order cyclic Jacobi on the whole matrix has
// N-1 rounds of N/2 disjoint rotations per sweep and one wavefront covers a round in two LDS round trips:
// lanes k < N/2 compute (c,s) of pair k; lane (k = lane&15, g = lane>>4) then rotates the 2x2 blocks
// rows{p,q} x cols{p2,q2} of pairs (k, k2 = g+4h) and rows g+4h' of the eigenvector columns p,q.
// Same warm start (A0 = V'AV from the previous call's V, Newton-Schulz re-orthogonalisation every kPsdWarmPeriod calls), same
// rotation formula, same packed layout.  Scratch per matrix: V (N x N) at woff, state at the end of the slot.
// ---------------------------------------------------------------------------
constexpr int kPsdSmallMax = 32;
constexpr int kPsdSLd = 33;

// dot product of the one-wavefront kernel's small GEMMs: the terms are added in index order (same bits as the plain loop), but four
// pairs of LDS reads are in flight at a time — a lone wavefront has nothing else to cover the LDS latency with (round 3: the four
// GEMM loops took 20 of the kernel's 48 us at order 20)
template <class FA, class FB>
__device__ __forceinline__ double psd_small_dot(int N, FA a_at, FB b_at) {
  double acc = 0.;
  int kk = 0;
  for (; kk + 4 <= N; kk += 4) {
    const double a0 = a_at(kk), a1 = a_at(kk + 1), a2 = a_at(kk + 2), a3 = a_at(kk + 3);
    const double b0 = b_at(kk), b1 = b_at(kk + 1), b2 = b_at(kk + 2), b3 = b_at(kk + 3);
    acc += a0 * b0;
    acc += a1 * b1;
    acc += a2 * b2;
    acc += a3 * b3;
  }
  for (; kk < N; ++kk) acc += a_at(kk) * b_at(kk);
  return acc;
}
__device__ __forceinline__ void d_proj_psd_small(double *x, PsdBatch B, double *scratch, int allow_warm,
                                                 const int *stall, const double *tol2) {
  SCS_STALL_GUARD(stall);
  const double offtol2 = psd_offtol2(tol2);
  __shared__ double S[32 * kPsdSLd], V[32 * kPsdSLd], T[32 * kPsdSLd];
  __shared__ double csc[16], css[16];
  const int lane = threadIdx.x, cidx = blockIdx.x;
  const int n = B.order[cidx];
  double *X = x + B.off[cidx];
  if (n == 0) return;
  if (n == 1) {
    if (lane == 0) X[0] = fmax(X[0], 0.);
    return;
  }
  const int N = (n + 1) & ~1, H = N / 2, ld = kPsdSLd;
  double *Vg = scratch + B.woff[cidx];
  double *state = Vg + psd_scratch_doubles(n) - kPsdStateDoubles;
  const double isq2 = 0.70710678118654752440, sq2 = 1.41421356237309504880;
  const bool warm = allow_warm && state[0] >= 1.;
  const bool reorth = warm && ((long)state[0] % kPsdWarmPeriod) == 0;
#if PSD_PROFILE
  double prof[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
#endif
  PSD_TICK(t_begin);

  for (int e = lane; e < N * N; e += 64) {
    const int j = e / N, i = e - j * N;
    S[i + ld * j] = 0.;
    V[i + ld * j] = warm ? Vg[e] : (i == j ? 1. : 0.);
  }
  wave_sync();
  for (int e = lane; e < n * n; e += 64) {
    const int j = e / n, i = e - j * n;
    if (i < j) continue;
    const long base = (long)j * n - (long)j * (j - 1) / 2;
    double v = X[base + (i - j)];
    if (i != j) v *= isq2;
    S[i + ld * j] = v;
    S[j + ld * i] = v;
  }
  wave_sync();
  PSD_TICK(t_unpacked);
  PSD_ACC(1, t_begin, t_unpacked);
  if (reorth) {  // V <- V (3I - V'V) / 2
    for (int e = lane; e < N * N; e += 64) {
      const int j = e / N, i = e - j * N;
      double acc = 0.;
      for (int k = 0; k < N; ++k) acc += V[k + ld * i] * V[k + ld * j];
      T[i + ld * j] = (i == j ? 1.5 : 0.) - 0.5 * acc;
    }
    wave_sync();
    double vn[(kPsdSmallMax * kPsdSmallMax + 63) / 64];
#pragma unroll
    for (int h = 0; h < (kPsdSmallMax * kPsdSmallMax + 63) / 64; ++h) {
      const int e = lane + 64 * h, j = e / N, i = e - j * N;
      double acc = 0.;
      if (e < N * N)
        for (int k = 0; k < N; ++k) acc += V[i + ld * k] * T[k + ld * j];
      vn[h] = acc;
    }
    wave_sync();
#pragma unroll
    for (int h = 0; h < (kPsdSmallMax * kPsdSmallMax + 63) / 64; ++h) {
      const int e = lane + 64 * h, j = e / N, i = e - j * N;
      if (e < N * N) V[i + ld * j] = vn[h];
    }
    wave_sync();
  }
  if (warm) {  // S <- V' S V
    for (int e = lane; e < N * N; e += 64) {
      const int j = e / N, i = e - j * N;
      T[i + ld * j] = psd_small_dot(N, [&](int k) { return S[i + ld * k]; }, [&](int k) { return V[k + ld * j]; });
    }
    wave_sync();
    for (int e = lane; e < N * N; e += 64) {
      const int j = e / N, i = e - j * N;
      if (i < j) continue;
      const double acc = psd_small_dot(N, [&](int k) { return V[k + ld * i]; }, [&](int k) { return T[k + ld * j]; });
      S[i + ld * j] = acc;
      S[j + ld * i] = acc;
    }
    wave_sync();
  }

  PSD_TICK(t_warmed);
  PSD_ACC(2, t_unpacked, t_warmed);
  const int k = lane & 15, g = lane >> 4;
  for (int sweep = 0; sweep < kPsdMaxSweeps; ++sweep) {
    double off = 0., tot = 0.;
    for (int e = lane; e < N * N; e += 64) {
      const int j = e / N, i = e - j * N;
      const double a = S[i + ld * j];
      tot += a * a;
      if (i != j) off += a * a;
    }
    off = wave_sum(off);
    tot = wave_sum(tot);
    off = __shfl(off, 0, 64);
    tot = __shfl(tot, 0, 64);
#if PSD_PROFILE >= 2
    if (lane == 0 && cidx == 0) printf("  small: before sweep %d  off_rel %.3e\n", sweep, sqrt(off / tot));
#endif
    if (off <= offtol2 * tot || off == 0.) break;
#if PSD_PROFILE
    prof[7] += 1.;
#endif
    for (int r = 0; r < N - 1; ++r) {
      // Round-robin pairing of round r, computed in registers.  The whole round is branch-free up to the
      // stores (idle lanes read entry 0): all LDS reads leave in one batch, i.e. ONE round trip before the
      // rotation and one (the published (c,s) of the other pairs) after it.
      auto pair_of = [&](int kk, int &pp, int &qq) {
        int x = r + kk, y = r - kk + (N - 1);
        x = x >= N - 1 ? x - (N - 1) : x;
        y = y >= N - 1 ? y - (N - 1) : y;
        x = kk == 0 ? N - 1 : x;
        y = kk == 0 ? r : y;
        pp = min(x, y);
        qq = max(x, y);
      };
      const bool kv = k < H;
      int p, q;
      pair_of(kv ? k : 0, p, q);
      const double apq = S[p + ld * q], app = S[p + ld * p], aqq = S[q + ld * q];
      int p2[4], q2[4];
      bool bv[4], rv[8];
      double a[4][4], vp[8], vq[8];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int k2 = g + 4 * h;
        bv[h] = kv && k2 < H;
        pair_of(bv[h] ? k2 : 0, p2[h], q2[h]);
        a[h][0] = S[p + ld * p2[h]]; a[h][1] = S[p + ld * q2[h]];
        a[h][2] = S[q + ld * p2[h]]; a[h][3] = S[q + ld * q2[h]];
      }
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int i = g + 4 * h;
        rv[h] = kv && i < N;
        const int ii = rv[h] ? i : 0;
        vp[h] = V[ii + ld * p];
        vq[h] = V[ii + ld * q];
      }
      {  // every lane forms the rotation of its own pair (lanes sharing k compute the same bits); lanes < H publish it
        const bool rot = fabs(apq) > 1e-300;
        double c, s;
        jacobi_rot(app, aqq, rot ? apq : 1.0, c, s);
        c = rot ? c : 1.;
        s = rot ? s : 0.;
        if (lane < H) {
          csc[lane] = c;
          css[lane] = s;
        }
      }
      wave_sync();  // (c,s) of every pair are in LDS; every lane has read the previous round's S and V
      const double c = csc[kv ? k : 0], s = css[kv ? k : 0];
      double c2[4], s2[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        c2[h] = csc[bv[h] ? g + 4 * h : 0];
        s2[h] = css[bv[h] ? g + 4 * h : 0];
      }
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const double t1 = c2[h] * a[h][0] - s2[h] * a[h][1], t2 = s2[h] * a[h][0] + c2[h] * a[h][1];
        const double t3 = c2[h] * a[h][2] - s2[h] * a[h][3], t4 = s2[h] * a[h][2] + c2[h] * a[h][3];
        if (bv[h]) {
          S[p + ld * p2[h]] = c * t1 - s * t3;
          S[p + ld * q2[h]] = c * t2 - s * t4;
          S[q + ld * p2[h]] = s * t1 + c * t3;
          S[q + ld * q2[h]] = s * t2 + c * t4;
        }
      }
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        if (rv[h]) {
          const int i = g + 4 * h;
          V[i + ld * p] = c * vp[h] - s * vq[h];
          V[i + ld * q] = s * vp[h] + c * vq[h];
        }
      }
      wave_sync();
    }
  }

  PSD_TICK(t_swept);
  PSD_ACC(3, t_warmed, t_swept);
  // X+ = V F V' with the second-order-accurate F of the block kernel's reconstruction (see there);
  // V itself goes back to the scratch for the next call's warm start
  for (int e = lane; e < N * N; e += 64) {
    const int j = e / N, i = e - j * N;
    Vg[e] = V[i + ld * j];
    const double di = i < n ? S[i + ld * i] : 0., dj = j < n ? S[j + ld * j] : 0.;
    double fij;
    if (i == j) {
      fij = fmax(di, 0.);
    } else {
      const double hi = fmax(di, dj), lo = fmin(di, dj);
      fij = S[i + ld * j] * (lo > 0. ? 1. : (hi <= 0. ? 0. : hi / (hi - lo)));
    }
    T[i + ld * j] = fij;  // F (every lane reads only diagonal entries of S it does not write: S is left untouched)
  }
  if (lane == 0) state[0] = warm ? state[0] + 1. : 1.;
  wave_sync();
  for (int e = lane; e < N * N; e += 64) {  // S <- V F
    const int j = e / N, i = e - j * N;
    S[i + ld * j] = psd_small_dot(N, [&](int kk) { return V[i + ld * kk]; }, [&](int kk) { return T[kk + ld * j]; });
  }
  wave_sync();
  for (int e = lane; e < n * n; e += 64) {  // X+ = (V F) V', lower triangle
    const int j = e / n, i = e - j * n;
    if (i < j) continue;
    const double acc = psd_small_dot(N, [&](int kk) { return S[i + ld * kk]; }, [&](int kk) { return V[j + ld * kk]; });
    const long base = (long)j * n - (long)j * (j - 1) / 2;
    X[base + (i - j)] = (i == j) ? acc : acc * sq2;
  }
#if PSD_PROFILE
  wave_sync();
  PSD_TICK(t_end);
  PSD_ACC(6, t_swept, t_end);
  if (lane == 0)
    for (int i = 1; i < 8; ++i) state[i] = prof[i];
#endif
}
__global__ __launch_bounds__(64) void k_proj_psd_small(double *x, PsdBatch B, double *scratch, int allow_warm,
                                                       const int *stall, const double *tol2) {
  d_proj_psd_small(x, B, scratch, allow_warm, stall, tol2);
}

// Round 4: the same projection by FOUR wavefronts (one per SIMD of a CU), with the round-robin done by MOVING the data.
// A lone wavefront spends a round of the sweep on its own latencies — 35 LDS reads, 128 fp64 operations at 4 clk each, 32 stores,
// index arithmetic for five pairs, ~1 us — and nothing overlaps them.  Here the matrix is kept by POSITION: the pairs of every
// round are the positions (2k, 2k+1), and after the rotations every row / column moves to the position the tournament gives it
// next (position 0 stays; 1 -> 2; odd 2k+1 -> 2k-1; even 2k -> 2k+2; the last even -> the last odd): the SAME permutation every
// round, so every address of the round loop is a per-lane constant.  Lane (k = tid & 15, g = tid >> 4) owns the 2x2 block
// (pair k, pair g) of S and rows g, g + 16 of the eigenvector columns of pair k, forms the rotations of BOTH its pairs itself (no
// published (c, s): no second barrier) and writes to the OTHER copy of S and V at the permuted positions — every entry is rewritten
// in every round, so the copies ping-pong: one barrier per round.  After N - 1 rounds (one sweep) everything is back at its home
// position, so the sweep test, the warm start and the reconstruction see the ordinary layout.  Same rotation formula, same
// sweep test and reconstruction as the one-wavefront kernel; the pairs meet in a different order, so the results agree with it
// to the sweep tolerance, not bit for bit (SCS_HIP_PSD_SMALL_WAVES=1 runs the one-wavefront kernel).
constexpr int kPsdSmallThreads = 256;
__device__ __forceinline__ int psd_small_next_pos(int a, int H) {
  if (H == 1 || a == 0) return a;
  const int kk = a >> 1;
  if (a & 1) return kk == 0 ? 2 : 2 * kk - 1;
  return kk == H - 1 ? 2 * H - 1 : 2 * kk + 2;
}
__device__ __forceinline__ void d_proj_psd_small4(double *x, PsdBatch B, double *scratch, int allow_warm,
                                                  const int *stall, const double *tol2, int cidx) {
  SCS_STALL_GUARD(stall);
  constexpr int NT = kPsdSmallThreads, SZ = 32 * kPsdSLd;
  const double offtol2 = psd_offtol2(tol2);
  __shared__ double SS[2 * SZ], VV[2 * SZ], T[SZ];
  const int tid = threadIdx.x;
  const int n = B.order[cidx];
  double *X = x + B.off[cidx];
  if (n == 0) return;
  if (n == 1) {
    if (tid == 0) X[0] = fmax(X[0], 0.);
    return;
  }
  const int N = (n + 1) & ~1, H = N / 2, ld = kPsdSLd;
  double *Vg = scratch + B.woff[cidx];
  double *state = Vg + psd_scratch_doubles(n) - kPsdStateDoubles;
  const double isq2 = 0.70710678118654752440, sq2 = 1.41421356237309504880;
  const double st0 = state[0];
  const bool warm = allow_warm && st0 >= 1.;
  const bool reorth = warm && ((long)st0 % kPsdWarmPeriod) == 0;
  int cur = 0;  // which copy of S and V is current
#if PSD_PROFILE
  double prof[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
#endif
  PSD_TICK(t_begin);
#define PSD_S(i, j) SS[cur * SZ + (i) + ld * (j)]
#define PSD_SN(i, j) SS[(cur ^ 1) * SZ + (i) + ld * (j)]
#define PSD_V(i, j) VV[cur * SZ + (i) + ld * (j)]

  for (int e = tid; e < N * N; e += NT) {
    const int j = e / N, i = e - j * N;
    PSD_S(i, j) = 0.;
    PSD_V(i, j) = warm ? Vg[e] : (i == j ? 1. : 0.);
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += NT) {
    const int j = e / n, i = e - j * n;
    if (i < j) continue;
    const long base = (long)j * n - (long)j * (j - 1) / 2;
    double v = X[base + (i - j)];
    if (i != j) v *= isq2;
    PSD_S(i, j) = v;
    PSD_S(j, i) = v;
  }
  __syncthreads();
  PSD_TICK(t_unpacked);
  PSD_ACC(1, t_begin, t_unpacked);
  if (reorth) {  // V <- V (3I - V'V) / 2
    for (int e = tid; e < N * N; e += NT) {
      const int j = e / N, i = e - j * N;
      double acc = 0.;
      for (int k = 0; k < N; ++k) acc += PSD_V(k, i) * PSD_V(k, j);
      T[i + ld * j] = (i == j ? 1.5 : 0.) - 0.5 * acc;
    }
    __syncthreads();
    double vn[(kPsdSmallMax * kPsdSmallMax + NT - 1) / NT];
#pragma unroll
    for (int h = 0; h < (kPsdSmallMax * kPsdSmallMax + NT - 1) / NT; ++h) {
      const int e = tid + NT * h, j = e / N, i = e - j * N;
      double acc = 0.;
      if (e < N * N)
        for (int k = 0; k < N; ++k) acc += PSD_V(i, k) * T[k + ld * j];
      vn[h] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < (kPsdSmallMax * kPsdSmallMax + NT - 1) / NT; ++h) {
      const int e = tid + NT * h, j = e / N, i = e - j * N;
      if (e < N * N) PSD_V(i, j) = vn[h];
    }
    __syncthreads();
  }
  if (warm) {  // S <- V' S V
    for (int e = tid; e < N * N; e += NT) {
      const int j = e / N, i = e - j * N;
      T[i + ld * j] = psd_small_dot(N, [&](int k) { return PSD_S(i, k); }, [&](int k) { return PSD_V(k, j); });
    }
    __syncthreads();
    for (int e = tid; e < N * N; e += NT) {
      const int j = e / N, i = e - j * N;
      if (i < j) continue;
      const double acc = psd_small_dot(N, [&](int k) { return PSD_V(k, i); }, [&](int k) { return T[k + ld * j]; });
      PSD_S(i, j) = acc;
      PSD_S(j, i) = acc;
    }
    __syncthreads();
  }

  PSD_TICK(t_warmed);
  PSD_ACC(2, t_unpacked, t_warmed);
  const int lane = tid & 63;
  // Per-lane constants of the round loop (offsets in doubles inside one copy).  Lanes without a pair / block / row work on
  // entry (0, 1) and write to a slot of the unused 33rd row, so that the round has no branch: the compiler then interleaves the
  // two rotation chains instead of running the second one inside the `if` of the stores.
  const int k = tid & 15, g = tid >> 4;
  const bool kv = k < H, bv = kv && g < H;
  const int p = kv ? 2 * k : 0, q = p + 1, p2 = bv ? 2 * g : 0, q2 = p2 + 1;
  const int np = psd_small_next_pos(p, H), nq = psd_small_next_pos(q, H);
  const int np2 = psd_small_next_pos(p2, H), nq2 = psd_small_next_pos(q2, H);
  const int dummy = 32 + ld * (tid & 31);
  const int o_pp = p + ld * p, o_qq = q + ld * q, o_pq = p + ld * q;       // diagonal block of pair k
  const int o_pp2 = p2 + ld * p2, o_qq2 = q2 + ld * q2, o_pq2 = p2 + ld * q2;  // ... of pair g
  const int o_a0 = p + ld * p2, o_a1 = p + ld * q2, o_a2 = q + ld * p2, o_a3 = q + ld * q2;
  const int w_a0 = bv ? np + ld * np2 : dummy, w_a1 = bv ? np + ld * nq2 : dummy;
  const int w_a2 = bv ? nq + ld * np2 : dummy, w_a3 = bv ? nq + ld * nq2 : dummy;
  const bool rv0 = kv && g < N, rv1 = kv && g + 16 < N;
  const int i0 = rv0 ? g : 0, i1 = rv1 ? g + 16 : 0;
  const int o_vp0 = i0 + ld * p, o_vq0 = i0 + ld * q, o_vp1 = i1 + ld * p, o_vq1 = i1 + ld * q;
  const int w_vp0 = rv0 ? i0 + ld * np : dummy, w_vq0 = rv0 ? i0 + ld * nq : dummy;
  const int w_vp1 = rv1 ? i1 + ld * np : dummy, w_vq1 = rv1 ? i1 + ld * nq : dummy;
  // one round from copy C to copy 1 - C (compile-time: the copy's offset folds into the LDS instructions)
  auto round = [&](auto CUR) {
    constexpr int C = decltype(CUR)::value;
    const double *Sr = SS + C * SZ, *Vr = VV + C * SZ;
    double *Sw = SS + (1 - C) * SZ, *Vw = VV + (1 - C) * SZ;
    const double apq = Sr[o_pq], app = Sr[o_pp], aqq = Sr[o_qq];
    const double bpq = Sr[o_pq2], bpp = Sr[o_pp2], bqq = Sr[o_qq2];
    const double a0 = Sr[o_a0], a1 = Sr[o_a1], a2 = Sr[o_a2], a3 = Sr[o_a3];
    const double vp0 = Vr[o_vp0], vq0 = Vr[o_vq0], vp1 = Vr[o_vp1], vq1 = Vr[o_vq1];
    double c, s, c2, s2;
    const bool rot = fabs(apq) > 1e-300, rot2 = fabs(bpq) > 1e-300;
    jacobi_rot(app, aqq, rot ? apq : 1.0, c, s);
    jacobi_rot(bpp, bqq, rot2 ? bpq : 1.0, c2, s2);
    c = rot ? c : 1.;
    s = rot ? s : 0.;
    c2 = rot2 ? c2 : 1.;
    s2 = rot2 ? s2 : 0.;
    const double t1 = c2 * a0 - s2 * a1, t2 = s2 * a0 + c2 * a1;
    const double t3 = c2 * a2 - s2 * a3, t4 = s2 * a2 + c2 * a3;
    Sw[w_a0] = c * t1 - s * t3;
    Sw[w_a1] = c * t2 - s * t4;
    Sw[w_a2] = s * t1 + c * t3;
    Sw[w_a3] = s * t2 + c * t4;
    Vw[w_vp0] = c * vp0 - s * vq0;
    Vw[w_vq0] = s * vp0 + c * vq0;
    Vw[w_vp1] = c * vp1 - s * vq1;
    Vw[w_vq1] = s * vp1 + c * vq1;
    __syncthreads();
  };
  auto sweep_rounds = [&](auto PAR) {  // the N - 1 (odd) rounds of a sweep that starts in copy PAR and ends in the other
    constexpr int P = decltype(PAR)::value;
    for (int r = 0; r + 2 < N; r += 2) {
      round(std::integral_constant<int, P>{});
      round(std::integral_constant<int, 1 - P>{});
    }
    round(std::integral_constant<int, P>{});
  };
  for (int sweep = 0; sweep < kPsdMaxSweeps; ++sweep) {
    double off = 0., tot = 0.;  // every wavefront forms the whole sum (same order, same bits: the exit below is uniform)
    for (int e = lane; e < N * N; e += 64) {
      const int j = e / N, i = e - j * N;
      const double a = PSD_S(i, j);
      tot += a * a;
      if (i != j) off += a * a;
    }
    off = wave_sum(off);
    tot = wave_sum(tot);
    off = __shfl(off, 0, 64);
    tot = __shfl(tot, 0, 64);
    if (off <= offtol2 * tot || off == 0.) break;
#if PSD_PROFILE
    prof[7] += 1.;
#endif
    if (cur == 0) sweep_rounds(std::integral_constant<int, 0>{});
    else sweep_rounds(std::integral_constant<int, 1>{});
    cur ^= 1;
  }

  PSD_TICK(t_swept);
  PSD_ACC(3, t_warmed, t_swept);
  for (int e = tid; e < N * N; e += NT) {
    const int j = e / N, i = e - j * N;
    Vg[e] = PSD_V(i, j);
    const double di = i < n ? PSD_S(i, i) : 0., dj = j < n ? PSD_S(j, j) : 0.;
    double fij;
    if (i == j) {
      fij = fmax(di, 0.);
    } else {
      const double hi = fmax(di, dj), lo = fmin(di, dj);
      fij = PSD_S(i, j) * (lo > 0. ? 1. : (hi <= 0. ? 0. : hi / (hi - lo)));
    }
    T[i + ld * j] = fij;
  }
  if (tid == 0) state[0] = warm ? st0 + 1. : 1.;
  __syncthreads();
  for (int e = tid; e < N * N; e += NT) {  // S(other copy) <- V F
    const int j = e / N, i = e - j * N;
    PSD_SN(i, j) = psd_small_dot(N, [&](int kk) { return PSD_V(i, kk); }, [&](int kk) { return T[kk + ld * j]; });
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += NT) {  // X+ = (V F) V', lower triangle
    const int j = e / n, i = e - j * n;
    if (i < j) continue;
    const double acc = psd_small_dot(N, [&](int kk) { return PSD_SN(i, kk); }, [&](int kk) { return PSD_V(j, kk); });
    const long base = (long)j * n - (long)j * (j - 1) / 2;
    X[base + (i - j)] = (i == j) ? acc : acc * sq2;
  }
#if PSD_PROFILE
  __syncthreads();
  PSD_TICK(t_end);
  PSD_ACC(6, t_swept, t_end);
  if (tid == 0)
    for (int i = 1; i < 8; ++i) state[i] = prof[i];
#endif
#undef PSD_S
#undef PSD_SN
#undef PSD_V
}
__global__ __launch_bounds__(kPsdSmallThreads) void k_proj_psd_small4(double *x, PsdBatch B, double *scratch, int allow_warm,
                                                                      const int *stall, const double *tol2) {
  d_proj_psd_small4(x, B, scratch, allow_warm, stall, tol2, (int)blockIdx.x);
}

// Short second-order cones and small PSD matrices in ONE launch (both project slices of the same vector in place and are independent
// of each other): workgroups [0, soc_blocks) run d_proj_soc_wave, the others one matrix each.  A lone config-5 problem's iteration is
// nine dependent launches of a few microseconds; this is one less (same bodies: same bits).
static_assert(kConeThreads == kPsdSmallThreads, "one launch geometry for both bodies");
__device__ __forceinline__ void d_proj_soc_psd_small(double *x, const int *__restrict__ soc_off, const int *__restrict__ soc_dim, int n_soc,
                                                     int soc_G, int soc_blocks, PsdBatch B, double *scratch, int allow_warm,
                                                     const int *stall, const double *tol2) {
  if ((int)blockIdx.x < soc_blocks) d_proj_soc_wave(x, soc_off, soc_dim, n_soc, soc_G, stall, (int)blockIdx.x);
  else d_proj_psd_small4(x, B, scratch, allow_warm, stall, tol2, (int)blockIdx.x - soc_blocks);
}
__global__ __launch_bounds__(kPsdSmallThreads) void k_proj_soc_psd_small(double *x, const int *__restrict__ soc_off,
                                                                         const int *__restrict__ soc_dim, int n_soc, int soc_G, int soc_blocks,
                                                                         PsdBatch B, double *scratch, int allow_warm, const int *stall,
                                                                         const double *tol2) {
  d_proj_soc_psd_small(x, soc_off, soc_dim, n_soc, soc_G, soc_blocks, B, scratch, allow_warm, stall, tol2);
}

// ---------------------------------------------------------------------------
// Complex PSD cone `cs` (R:scs/scsobject.h:734-737; k*k reals per order-k Hermitian matrix,
// R:test/test_spectral_and_complex_cones.py:22-24, R:test/test_mix_sd_csd_cone.py:34-35).
// Element order inside the slice (UPSTREAM-RECALL, not evidenced in the reference): lower triangle,
// column by column: H_jj, then (sqrt2 Re H_ij, sqrt2 Im H_ij) for i > j.
// H = A + iB is PSD iff the real symmetric M = [[A, -B], [B, A]] is, and Pi(M) = [[A+, -B+], [B+, A+]]:
// the slice is expanded into the packed vector of the 2k x 2k embedding, projected by k_proj_psd
// (same MFMA kernel, same warm start), and read back (averaging the two copies of every entry).
// Both layouts carry the sqrt(2) on off-diagonals, so the expansion is a signed copy.
// ---------------------------------------------------------------------------
struct CsBatch {
  const int *off;     // start of each cone's k*k slice inside the m-vector slice
  const int *order;   // k
  const long *soff;   // start of the packed 2k x 2k embedding inside the staging buffer
  int count;
};

__device__ __forceinline__ long cs_col_start(long j, long k) { return j * (2 * k - j); }
__device__ __forceinline__ long packed_idx(long I, long J, long N) { return J * N - J * (J - 1) / 2 + (I - J); }  // I >= J

__global__ __launch_bounds__(256) void k_cs_expand(const double *__restrict__ x, CsBatch B, double *stage, const int *stall) {
  SCS_STALL_GUARD(stall);
  const int c = blockIdx.x;
  const long k = B.order[c], N = 2 * k;
  const double *X = x + B.off[c];
  double *P = stage + B.soff[c];
  for (long e = threadIdx.x; e < N * N; e += blockDim.x) {
    const long J = e / N, I = e % N;
    if (I < J) continue;
    const long i = I < k ? I : I - k, j = J < k ? J : J - k;
    double v;
    if ((I < k) == (J < k)) {  // diagonal blocks: A (i >= j here)
      v = (i == j) ? X[cs_col_start(j, k)] : X[cs_col_start(j, k) + 1 + 2 * (i - j - 1)];
    } else {  // lower-left block: B_ij, antisymmetric
      if (i == j) v = 0.;
      else if (i > j) v = X[cs_col_start(j, k) + 2 + 2 * (i - j - 1)];
      else v = -X[cs_col_start(i, k) + 2 + 2 * (j - i - 1)];
    }
    P[packed_idx(I, J, N)] = v;
  }
}

__global__ __launch_bounds__(256) void k_cs_extract(double *x, CsBatch B, const double *__restrict__ stage, const int *stall) {
  SCS_STALL_GUARD(stall);
  const int c = blockIdx.x;
  const long k = B.order[c], N = 2 * k;
  double *X = x + B.off[c];
  const double *P = stage + B.soff[c];
  for (long e = threadIdx.x; e < k * k; e += blockDim.x) {
    const long j = e / k, i = e % k;
    if (i < j) continue;
    const long cj = cs_col_start(j, k);
    if (i == j) {
      X[cj] = 0.5 * (P[packed_idx(j, j, N)] + P[packed_idx(k + j, k + j, N)]);
    } else {
      X[cj + 1 + 2 * (i - j - 1)] = 0.5 * (P[packed_idx(i, j, N)] + P[packed_idx(k + i, k + j, N)]);
      X[cj + 2 + 2 * (i - j - 1)] = 0.5 * (P[packed_idx(k + i, j, N)] - P[packed_idx(k + j, i, N)]);
    }
  }
}

}  // namespace scship
