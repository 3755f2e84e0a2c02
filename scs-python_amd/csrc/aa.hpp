// aa.hpp — K10: Anderson acceleration on a device-resident iterate (length dim = n+m+1 inside the ADMM loop).
//
// Plays the role of scs_source/src/aa.c (named at R:meson.build:187; absent): same interface
// (init / apply / safeguard / reset / finish), knobs R:README.md:98-104, statistics R:scs/scsobject.h:1096-1107.
//
// Data layout: S, Y, D are tall-skinny dim x mem column-major matrices in HBM (3 * dim * mem * 8 B; 720 MB at
// dim = 3e6, mem = 10 — trivial against 288 GB).  One call = one streaming pass that writes the new column of S, Y, D
// (k_aa_update), then the least-squares weights
//     type-I : gamma = (S'Y + r I)^{-1} S'g        type-II: gamma = (Y'Y + r I)^{-1} Y'g ,   r = regularization * ||M||_F
// and the extrapolation f -= D gamma (k_aa_apply).  Two ways to the weights:
//
//  * TSQR (default): ONE pass over the history W = [L | Y | g] (L = S or Y) factors it with Householder reflectors,
//    tile by tile, one wavefront per tile chain: the tile (64 * rho rows x c columns) and the running triangle live in
//    LDS, the reflector of pivot column k is applied to the remaining columns in batches of four whose dot products are
//    reduced with interleaved DPP trees.  Only the L columns are pivots (the rows of [R_LL | Q'Y | Q'g] are all the
//    system needs: L'Y = R_LL'(Q'Y), L'g = R_LL'(Q'g)), the per-wave triangles are stacked and reduced by the same
//    kernel in a fixed tree (1024 -> 16 -> 1 waves), and a one-wave kernel forms the regularised mem x mem system from
//    the final triangle, solves it by LU with partial pivoting, applies the rank / finite / weight-cap tests and leaves
//    gamma + verdict in device memory: nothing but 8 + mem doubles crosses PCIe.  Fixed order everywhere => bitwise
//    run-to-run determinism.
//  * Gram (SCS_HIP_AA=gram; also the A/B reference of tests/test_aa_gpu.py): per call one column changes, so the
//    system matrix is updated incrementally from one fused pass (new row, new column, L'g: k_aa_dots) and solved on
//    the host.
// Both are compared step by step with the CPU checker in tests/test_aa_gpu.py.
#pragma once
#include "common.hpp"
#include "vec.hpp"

namespace scship {

constexpr int kAaMaxMem = 32;  // columns per register block; the TSQR path takes histories up to this length
constexpr int kAaMaxCols = 2 * kAaMaxMem + 1;

// device result record of one solve (doubles): verdict, statistics, then gamma
enum : int { AA_R_OK = 0, AA_R_RANK, AA_R_CODE, AA_R_NORM, AA_R_REG, AA_R_NORMG, AA_R_NORMD, AA_R_GAMMA = 8, AA_R_COUNT = 8 + kAaMaxMem };
enum : int { AA_CODE_OK = 0, AA_CODE_RANK0, AA_CODE_LAPACK, AA_CODE_NONFINITE, AA_CODE_WEIGHT };

// first call after a reset: x_prev = x, f_prev = f, g_prev = x - f
__device__ __forceinline__ void d_aa_seed(const double *__restrict__ x, const double *__restrict__ f,
                                          double *ax, double *af, double *gprev, long dim) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    const double xi = x[i], fi = f[i];
    ax[i] = xi;
    af[i] = fi;
    gprev[i] = xi - fi;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_aa_seed(const double *__restrict__ x,
                                                         const double *__restrict__ f, double *ax, double *af,
                                                         double *gprev, long dim) {
  d_aa_seed(x, f, ax, af, gprev, dim);
}

// g = x - f; s = x - x_prev; d = f - f_prev; y = g - g_prev; store column idx of S,D,Y;
// roll x_prev,f_prev,g_prev; partial ||g||^2
__device__ __forceinline__ void d_aa_update(const double *__restrict__ x, const double *__restrict__ f,
                                            double *ax, double *af, double *gprev, double *S, double *Y,
                                            double *D, long dim, int idx, double *part) {
  __shared__ double sm[kVecThreads / 64];
  double acc = 0.;
  double *Sc = S + (size_t)dim * idx, *Yc = Y + (size_t)dim * idx, *Dc = D + (size_t)dim * idx;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    const double xi = x[i], fi = f[i];
    const double g = xi - fi;
    Sc[i] = xi - ax[i];
    Dc[i] = fi - af[i];
    Yc[i] = g - gprev[i];
    ax[i] = xi;
    af[i] = fi;
    gprev[i] = g;
    acc += g * g;
  }
  acc = block_sum<kVecThreads>(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(kVecThreads) void k_aa_update(const double *__restrict__ x,
                                                           const double *__restrict__ f, double *ax,
                                                           double *af, double *gprev, double *S, double *Y,
                                                           double *D, long dim, int idx, double *part) {
  d_aa_update(x, f, ax, af, gprev, S, Y, D, dim, idx, part);
}

// ------------------------------------------------------------------------------------------------ Gram path
// one streaming pass over L (= S or Y) and Y:  row[j] = L_idx . Y_j,  col[j] = L_j . Y_idx,  w[j] = L_j . g  for the
// columns j0 <= j < min(j0 + kAaMaxMem, len) (histories longer than kAaMaxMem columns take one launch per block of
// columns).  partial layout: part[(k*cap + j) * nb + b], k = 0 (row), 1 (col), 2 (w), cap = the history capacity
__global__ __launch_bounds__(kVecThreads) void k_aa_dots(const double *__restrict__ L, const double *__restrict__ Y,
                                                         const double *__restrict__ g, long dim, int len, int idx, double *part,
                                                         int j0, int cap) {
  __shared__ double sm[kVecThreads / 64];
  double row[kAaMaxMem], col[kAaMaxMem], w[kAaMaxMem];
#pragma unroll
  for (int j = 0; j < kAaMaxMem; ++j) row[j] = col[j] = w[j] = 0.;
  const double *Li = L + (size_t)dim * idx, *Yi = Y + (size_t)dim * idx;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    const double li = Li[i], yi = Yi[i], gi = g[i];
#pragma unroll
    for (int j = 0; j < kAaMaxMem; ++j) {
      if (j0 + j < len) {
        const double Lj = L[(size_t)dim * (j0 + j) + i], Yj = Y[(size_t)dim * (j0 + j) + i];
        row[j] += li * Yj;
        col[j] += Lj * yi;
        w[j] += Lj * gi;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kAaMaxMem; ++j) {
    if (j0 + j < len) {  // len is uniform across the grid
      const double a = block_sum<kVecThreads>(row[j], sm);
      const double b = block_sum<kVecThreads>(col[j], sm);
      const double c = block_sum<kVecThreads>(w[j], sm);
      if (threadIdx.x == 0) {
        part[((size_t)(0 * cap + j0 + j)) * gridDim.x + blockIdx.x] = a;
        part[((size_t)(1 * cap + j0 + j)) * gridDim.x + blockIdx.x] = b;
        part[((size_t)(2 * cap + j0 + j)) * gridDim.x + blockIdx.x] = c;
      }
    }
  }
}

// out[0] = sum of norm partials (||g||^2); out[1 + k*cap + j] = reduced dots; res[AA_R_NORMG] = ||g||
__global__ __launch_bounds__(kVecThreads) void k_fin_aa(const double *npart, int nnp, const double *part, int np, int len,
                                                        double *out, double *res, int cap) {
  __shared__ double sm[kVecThreads / 64];
  const double ng = part_sum(npart, nnp, sm);
  if (threadIdx.x == 0) {
    out[0] = ng;
    res[AA_R_NORMG] = sqrt(ng);
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k)
    for (int j = 0; j < len; ++j) {
      const double v = part_sum(part + ((size_t)(k * cap + j)) * np, np, sm);
      if (threadIdx.x == 0) out[1 + k * cap + j] = v;
      __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ TSQR path
// The tall matrix handed to one level of the reduction.  Level 1: the history itself — columns [L_0 .. L_{nL-1} |
// Y_0 .. Y_{nY-1} | g], each a contiguous vector of `rows` doubles (L, Y with leading dimension ld).  Levels >= 2:
// the stacked triangles of the level below, one column-major buffer (L = buffer, nL = c, ld = its row count).
struct AaTall {
  const double *L, *Y, *g;
  long ld, rows;
  int nL, nY, c, npiv;  // c columns in total, the first npiv are pivots
};
__device__ __forceinline__ const double *aa_col(const AaTall &W, int j) {
  if (j < W.nL) return W.L + (size_t)W.ld * j;
  if (j < W.nL + W.nY) return W.Y + (size_t)W.ld * (j - W.nL);
  return W.g;
}

// value of the DPP source lane, +0.0 where the source is outside the row or the lane is masked off
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double aa_dpp0(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
// NR independent 64-lane sums, their DPP trees interleaved (row_shr 1,2,3,4,8, row_bcast 15, 31: total in lane 63,
// handed to every lane through two v_readlane).  Fixed order: bitwise reproducible.
template <int NR>
__device__ __forceinline__ void aa_wave_allsum(double (&v)[NR]) {
  double s[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] = v[i] + aa_dpp0<0x111, 0xf, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] += aa_dpp0<0x112, 0xf, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] += aa_dpp0<0x113, 0xf, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] += aa_dpp0<0x114, 0xf, 0xe>(s[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] += aa_dpp0<0x118, 0xf, 0xc>(s[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] += aa_dpp0<0x142, 0xa, 0xf>(s[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] += aa_dpp0<0x143, 0xc, 0xf>(s[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(s[i]), 63), hi = __builtin_amdgcn_readlane(__double2hiint(s[i]), 63);
    v[i] = __hiloint2double(hi, lo);
  }
}

constexpr int kAaRhoMax = 4;  // a tile has 64 * rho rows (rho rows per lane)
inline int aa_pick_rho(int c) {  // ~40 KB of LDS per wavefront: 4 resident tile chains per CU
  const int r = (40 * 1024) / (64 * c * 8);
  return r < 1 ? 1 : (r > kAaRhoMax ? kAaRhoMax : r);
}
inline size_t aa_tsqr_lds(int c, int npiv, int rho) { return ((size_t)c * 64 * rho + (size_t)npiv * c) * sizeof(double); }

// One wavefront per workgroup.  Wave w factors tiles [w * tiles_per_wave, (w+1) * tiles_per_wave) into one running
// npiv x c triangle E (rows of [R_LL | Q'(other columns)]), which it stores as rows [w * npiv, (w+1) * npiv) of the
// column-major output (leading dimension out_ld).  Householder step k on the stacked [E row k; tile]: only E row k and
// the tile carry column k below the diagonal (E is upper trapezoidal), v = [alpha - beta; tile column k].
__device__ __forceinline__ void d_aa_tsqr(AaTall W, int rho, long tiles_per_wave, double *out, long out_ld) {
  extern __shared__ double aa_lds[];
  const int lane = threadIdx.x, c = W.c, npiv = W.npiv, TR = 64 * rho;
  double *T = aa_lds;                      // T[j * TR + r]: tile, column-major (lane-consecutive rows: conflict-free)
  double *E = aa_lds + (size_t)c * TR;     // E[i * c + j]
  for (int t = lane; t < npiv * c; t += 64) E[t] = 0.;
  const long ntiles = (W.rows + TR - 1) / TR;
  const long t0 = (long)blockIdx.x * tiles_per_wave, t1 = t0 + tiles_per_wave < ntiles ? t0 + tiles_per_wave : ntiles;
  for (long t = t0; t < t1; ++t) {
    const long r0 = t * TR;
    for (int j = 0; j < c; ++j) {
      const double *col = aa_col(W, j);
#pragma unroll
      for (int q = 0; q < kAaRhoMax; ++q)
        if (q < rho) {
          const long r = r0 + q * 64 + lane;
          T[(size_t)j * TR + q * 64 + lane] = r < W.rows ? col[r] : 0.;
        }
    }
    __syncthreads();  // (one wave: orders lane 0's writes of E against the other lanes' reads)
    for (int k = 0; k < npiv; ++k) {
      double xk[kAaRhoMax], ss[1] = {0.};
#pragma unroll
      for (int q = 0; q < kAaRhoMax; ++q) {
        xk[q] = q < rho ? T[(size_t)k * TR + q * 64 + lane] : 0.;
        ss[0] += xk[q] * xk[q];
      }
      aa_wave_allsum<1>(ss);
      const double sigma = ss[0], alpha = E[k * c + k];
      if (sigma == 0.) continue;  // nothing below the diagonal (wave-uniform; a NaN flows on into E)
      const double nrm = sqrt(alpha * alpha + sigma);
      const double beta = alpha >= 0. ? -nrm : nrm;
      const double vp = alpha - beta;
      const double tau = 1.0 / (nrm * (nrm + fabs(alpha)));  // 2 / v'v
      for (int j = k + 1; j < c; j += 4) {
        double tj[4][kAaRhoMax], d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[i] = 0.;
#pragma unroll
          for (int q = 0; q < kAaRhoMax; ++q) {
            tj[i][q] = (j + i < c && q < rho) ? T[(size_t)(j + i) * TR + q * 64 + lane] : 0.;
            d[i] += xk[q] * tj[i][q];
          }
        }
        aa_wave_allsum<4>(d);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (j + i < c) {
            const double ekj = E[k * c + j + i];
            const double wj = tau * (vp * ekj + d[i]);
#pragma unroll
            for (int q = 0; q < kAaRhoMax; ++q)
              if (q < rho) T[(size_t)(j + i) * TR + q * 64 + lane] = tj[i][q] - wj * xk[q];
            if (lane == 0) E[k * c + j + i] = ekj - wj * vp;
          }
        }
      }
      if (lane == 0) E[k * c + k] = beta;
    }
    __syncthreads();
  }
  for (int t = lane; t < npiv * c; t += 64) {
    const int i = t / c, j = t - i * c;
    out[(size_t)j * out_ld + (size_t)blockIdx.x * npiv + i] = E[t];
  }
}
__global__ __launch_bounds__(64) void k_aa_tsqr(AaTall W, int rho, long tiles_per_wave, double *out,
                                                long out_ld) {
  d_aa_tsqr(W, rho, tiles_per_wave, out, out_ld);
}

// ---- level 1 of the reduction for the default histories (lookback 10: c = 21 type-I, c = 11 type-II), round 3 ----
// Same Householder recurrences, tile by tile, as d_aa_tsqr — but the tile lives in REGISTERS (RHO rows per lane x C columns: no LDS
// traffic for it, three wavefronts per SIMD instead of one), a workgroup carries four independent chains (one per wavefront), and the
// C - k dot products of pivot k — the column's own norm included — are reduced TOGETHER by a reduce-scatter over the 64 lanes
// (exchange half of the values with the lane 32 away, half of the rest with the lane 16 away, ...: ~N exchanges for N values instead
// of a 7-step tree per value), then handed to every lane through 16 doubles of LDS.  Fixed exchange order: bitwise reproducible.
// (level-1 launch at l = 3e6, c = 21: 1.28 ms with d_aa_tsqr -> see profiles/r03_aa_tsqr.txt)
__device__ __forceinline__ void aa_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// v[0 .. NV) per lane -> v[0] = the 64-lane total of value number aa_rs_index<NV>(lane); NV a power of two <= 32.
// An exchange with the lane `m` away is two ds_bpermute_b32 (byte index (lane ^ m) * 4, computed once per kernel: no
// per-exchange index arithmetic — the kernel is VALU-bound, profiles/r03_aa_tsqr.txt) + two selects + one add.
struct AaXor {
  int idx[6];  // byte index of lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1
  __device__ __forceinline__ explicit AaXor(int lane) {
#pragma unroll
    for (int b = 0; b < 6; ++b) idx[b] = (lane ^ (32 >> b)) << 2;
  }
};
__device__ __forceinline__ double aa_xchg(double v, int byte_idx) {
  const int lo = __builtin_amdgcn_ds_bpermute(byte_idx, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(byte_idx, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// (compile-time recursion: the live count N and the step B must be constants, or the register array is indexed dynamically)
template <int NV, int N, int B>
__device__ __forceinline__ void aa_rs_step(double (&v)[NV], int lane, const AaXor &X) {
  if constexpr (B < 6) {
    constexpr int m = 32 >> B;
    if constexpr (N > 1) {
      constexpr int half = N / 2;
      const bool up = (lane & m) != 0;
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const double send = up ? v[i] : v[i + half], keep = up ? v[i + half] : v[i];
        v[i] = keep + aa_xchg(send, X.idx[B]);
      }
      aa_rs_step<NV, half, B + 1>(v, lane, X);
    } else {
      v[0] += aa_xchg(v[0], X.idx[B]);
      aa_rs_step<NV, 1, B + 1>(v, lane, X);
    }
  }
}
template <int NV>
__device__ __forceinline__ void aa_reduce_scatter(double (&v)[NV], int lane, const AaXor &X) {
  aa_rs_step<NV, NV, 0>(v, lane, X);
}
template <int N, int B>
__device__ __forceinline__ int aa_rs_index_step(int lane) {
  if constexpr (B < 6 && N > 1) return ((lane & (32 >> B)) ? N / 2 : 0) + aa_rs_index_step<N / 2, B + 1>(lane);
  else return 0;
}
template <int NV>
__device__ __forceinline__ int aa_rs_index(int lane) {
  return aa_rs_index_step<NV, 0>(lane);
}
#ifndef AA_FAST_RHO
#define AA_FAST_RHO 3
#endif
constexpr int kAaFastWaves = 4;  // chains per workgroup
constexpr int kAaFastRho = AA_FAST_RHO;  // rows per lane and tile
inline int aa_tsqr_fast_kind(int c, int npiv) { return (c == 21 && npiv == 10) ? 1 : (c == 11 && npiv == 10) ? 2 : 0; }

template <int C, int NPIV, int K>
__device__ __forceinline__ void aa_fast_pivot(double (&T)[C][kAaFastRho], double *E, double *bc, int lane, const AaXor &X) {
  constexpr int RHO = kAaFastRho, NVAL = C - K;                 // values of this pivot: sigma, then the dots with columns K + 1 ..
  constexpr int N0 = NVAL > 16 ? 16 : (NVAL > 8 ? 16 : (NVAL > 4 ? 8 : (NVAL > 2 ? 4 : 2)));
  constexpr int REST = NVAL > 16 ? NVAL - 16 : 0;
  constexpr int N1 = REST > 8 ? 16 : (REST > 4 ? 8 : (REST > 2 ? 4 : (REST > 0 ? 2 : 0)));
  {
    double d[N0];
#pragma unroll
    for (int i = 0; i < N0; ++i) {
      d[i] = 0.;
      if (i < NVAL) {
#pragma unroll
        for (int q = 0; q < RHO; ++q) d[i] += T[K][q] * T[K + i][q];
      }
    }
    aa_reduce_scatter<N0>(d, lane, X);
    bc[aa_rs_index<N0>(lane)] = d[0];
  }
  if constexpr (N1 > 0) {
    double d[N1];
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      d[i] = 0.;
      if (16 + i < NVAL) {
#pragma unroll
        for (int q = 0; q < RHO; ++q) d[i] += T[K][q] * T[K + 16 + i][q];
      }
    }
    aa_reduce_scatter<N1>(d, lane, X);
    bc[16 + aa_rs_index<N1>(lane)] = d[0];
  }
  aa_wave_sync();
  const double sigma = bc[0], alpha = E[K * C + K];
  if (sigma != 0.) {  // (wave-uniform; nothing below the diagonal otherwise — a NaN flows on into E)
    const double nrm = sqrt(alpha * alpha + sigma);
    const double beta = alpha >= 0. ? -nrm : nrm;
    const double vp = alpha - beta;
    const double tau = 1.0 / (nrm * (nrm + fabs(alpha)));  // 2 / v'v
#pragma unroll
    for (int i = 1; i < NVAL; ++i) {
      const double ekj = E[K * C + K + i];
      const double wj = tau * (vp * ekj + bc[i]);
#pragma unroll
      for (int q = 0; q < RHO; ++q) T[K + i][q] -= wj * T[K][q];
      if (lane == 0) E[K * C + K + i] = ekj - wj * vp;
    }
    if (lane == 0) E[K * C + K] = beta;
  }
  aa_wave_sync();  // bc is rewritten by the next pivot, E row K + 1 read by it
  if constexpr (K + 1 < NPIV) aa_fast_pivot<C, NPIV, K + 1>(T, E, bc, lane, X);
}

template <int C, int NPIV>
__device__ __forceinline__ void d_aa_tsqr_fast(AaTall W, long tiles_per_wave, long nw, double *out, long out_ld) {
  constexpr int RHO = kAaFastRho, TR = 64 * RHO;
  __shared__ double lds[kAaFastWaves][NPIV * C + 32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * kAaFastWaves + wv;
  if (gw >= nw) return;
  double *E = lds[wv], *bc = E + NPIV * C;
  for (int t = lane; t < NPIV * C; t += 64) E[t] = 0.;
  aa_wave_sync();
  const AaXor X(lane);
  const long ntiles = (W.rows + TR - 1) / TR;
  const long t0 = gw * tiles_per_wave, t1 = t0 + tiles_per_wave < ntiles ? t0 + tiles_per_wave : ntiles;
  for (long t = t0; t < t1; ++t) {
    const long r0 = t * TR;
    double T[C][RHO];
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const double *col = aa_col(W, j);
#pragma unroll
      for (int q = 0; q < RHO; ++q) {
        const long r = r0 + q * 64 + lane;
        T[j][q] = r < W.rows ? col[r] : 0.;
      }
    }
    aa_fast_pivot<C, NPIV, 0>(T, E, bc, lane, X);
  }
  for (int t = lane; t < NPIV * C; t += 64) {
    const int i = t / C, j = t - i * C;
    out[(size_t)j * out_ld + (size_t)gw * NPIV + i] = E[t];
  }
}
template <int C, int NPIV>
__global__ __launch_bounds__(64 * kAaFastWaves) void k_aa_tsqr_fast(AaTall W, long tiles_per_wave, long nw, double *out, long out_ld) {
  d_aa_tsqr_fast<C, NPIV>(W, tiles_per_wave, nw, out, out_ld);
}

// One wavefront: the regularised len x len system from the final triangle R (column-major npiv x c, npiv = len),
//   type-I : M = R_LL' R_LY (columns len .. 2 len - 1), w = R_LL' R_Lg;   type-II: M = R_YY' R_YY, w = R_YY' R_Yg,
// LU with partial pivoting (row swaps, then eliminations, the order DeviceAa::dense_solve uses), then the rank / finite /
// weight-cap tests of aa.c's solve.  Leaves the verdict and gamma in res.
__device__ __forceinline__ void d_aa_solve(const double *__restrict__ R, int len, int c, int type1,
                                           double regularization, double max_weight_norm, const double *npart,
                                           int nnp, double *res) {
  __shared__ double M[kAaMaxMem * kAaMaxMem], w[kAaMaxMem];
  const int lane = threadIdx.x;
  {
    double s = 0.;
    for (int i = lane; i < nnp; i += 64) s += npart[i];
    s = wave_sum(s);
    if (lane == 0) res[AA_R_NORMG] = sqrt(s);
  }
  const int yoff = type1 ? len : 0;
  for (int e = lane; e < len * len; e += 64) {
    const int i = e % len, j = e / len;
    double s = 0.;
    for (int k = 0; k < len; ++k) s += R[(size_t)i * len + k] * R[(size_t)(yoff + j) * len + k];
    M[i + len * j] = s;
  }
  for (int j = lane; j < len; j += 64) {
    double s = 0.;
    for (int k = 0; k < len; ++k) s += R[(size_t)j * len + k] * R[(size_t)(c - 1) * len + k];
    w[j] = s;
  }
  __syncthreads();
  __shared__ double reg_s;
  __shared__ int piv_s, sing_s;
  if (lane == 0) {
    double nrm = 0.;
    for (int j = 0; j < len; ++j)
      for (int i = 0; i < len; ++i) nrm += M[i + len * j] * M[i + len * j];
    reg_s = regularization * sqrt(nrm);
    sing_s = 0;
  }
  __syncthreads();
  const double reg = reg_s;
  if (regularization > 0 && lane < len) M[lane + len * lane] += reg;
  __syncthreads();
  // LU with partial pivoting.  Round 4: the row swap and the elimination of a step run across the lanes (every entry still goes
  // through the same operations in the same order: same bits as the one-lane loop, which took 36 us for a 10 x 10 system — a
  // chain of dependent LDS read-modify-writes); the pivot scan keeps its sequential first-maximum / NaN semantics on lane 0.
  int rank = 0;
  for (int k = 0; k < len; ++k) {
    if (lane == 0) {
      int piv = k;
      double mx = fabs(M[k + len * k]);
      for (int i = k + 1; i < len; ++i) {
        const double a = fabs(M[i + len * k]);
        if (a > mx) { mx = a; piv = i; }
      }
      piv_s = piv;
      if (!(mx > 0.) || !isfinite(mx)) sing_s = 1;
    }
    __syncthreads();
    if (sing_s) break;
    rank++;
    const int piv = piv_s;
    if (piv != k) {
      for (int j = lane; j <= len; j += 64) {  // column `len` = the right-hand side
        double *a = j < len ? &M[k + len * j] : &w[k], *b = j < len ? &M[piv + len * j] : &w[piv];
        const double t = *a;
        *a = *b;
        *b = t;
      }
      __syncthreads();
    }
    const int nr = len - 1 - k, nc = len - k;  // rows k+1 .. len-1, columns k+1 .. len-1 and the right-hand side
    const double dkk = M[k + len * k];
    for (int e = lane; e < nr * nc; e += 64) {
      const int i = k + 1 + e % nr, jj = e / nr, j = k + 1 + jj;
      const double f = M[i + len * k] / dkk;
      if (f == 0.) continue;
      if (j < len) M[i + len * j] -= f * M[k + len * j];
      else w[i] -= f * w[k];
    }
    __syncthreads();
  }
  if (lane == 0) {
    const bool singular = sing_s != 0;
    (void)singular;
    int code = AA_CODE_OK;
    double nw = 0.;
    if (rank == 0) code = AA_CODE_RANK0;
    else if (rank < len) code = AA_CODE_LAPACK;
    else {
      for (int k = len - 1; k >= 0; --k) {
        double s = w[k];
        for (int j = k + 1; j < len; ++j) s -= M[k + len * j] * w[j];
        w[k] = s / M[k + len * k];
      }
      for (int j = 0; j < len; ++j) nw += w[j] * w[j];
      nw = sqrt(nw);
      if (!isfinite(nw)) code = AA_CODE_NONFINITE;
      else if (nw >= max_weight_norm) code = AA_CODE_WEIGHT;
    }
    res[AA_R_OK] = code == AA_CODE_OK ? 1.0 : 0.0;
    res[AA_R_RANK] = (double)rank;
    res[AA_R_CODE] = (double)code;
    res[AA_R_NORM] = nw;
    res[AA_R_REG] = reg;
    for (int j = 0; j < len; ++j) res[AA_R_GAMMA + j] = code == AA_CODE_OK ? w[j] : 0.0;
  }
}
__global__ __launch_bounds__(64) void k_aa_solve(const double *__restrict__ R, int len, int c, int type1,
                                                 double regularization, double max_weight_norm,
                                                 const double *npart, int nnp, double *res) {
  d_aa_solve(R, len, c, type1, regularization, max_weight_norm, npart, nnp, res);
}

// f -= D gamma;  optional relaxation: f = beta f + (1-beta) (x - S gamma).  ok != nullptr: the device-side verdict
// of k_aa_solve (gamma sits right behind it); a rejected step leaves f untouched.  Any history length: the weights are
// read as they are used (uniform addresses: scalar loads), the columns are subtracted in ascending order.
__device__ __forceinline__ void d_aa_apply(double *f, const double *__restrict__ D, const double *__restrict__ S,
                                           const double *__restrict__ xcur, const double *__restrict__ gamma, long dim, int len,
                                           double relaxation, const double *ok) {
  if (ok && !(*ok != 0.)) return;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    double fi = f[i], xs = 0.;
    for (int j = 0; j < len; ++j) fi -= gamma[j] * D[(size_t)dim * j + i];
    if (relaxation != 1.0) {
      for (int j = 0; j < len; ++j) xs += gamma[j] * S[(size_t)dim * j + i];
      fi = relaxation * fi + (1. - relaxation) * (xcur[i] - xs);
    }
    f[i] = fi;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_aa_apply(double *f, const double *__restrict__ D,
                                                          const double *__restrict__ S,
                                                          const double *__restrict__ xcur,
                                                          const double *__restrict__ gamma, long dim, int len,
                                                          double relaxation, const double *ok) {
  d_aa_apply(f, D, S, xcur, gamma, dim, len, relaxation, ok);
}

// safeguard: partial ||x_new - f_new||^2
__device__ __forceinline__ void d_aa_diffsq(const double *__restrict__ a, const double *__restrict__ b,
                                            long dim, double *part) {
  __shared__ double sm[kVecThreads / 64];
  double acc = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    const double d = a[i] - b[i];
    acc += d * d;
  }
  acc = block_sum<kVecThreads>(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(kVecThreads) void k_aa_diffsq(const double *__restrict__ a,
                                                           const double *__restrict__ b, long dim,
                                                           double *part) {
  d_aa_diffsq(a, b, dim, part);
}
__device__ __forceinline__ void d_fin_safeguard(const double *part, int np, double factor, double *res,
                                                int *bad) {
  __shared__ double sm[kVecThreads / 64];
  const double s = part_sum(part, np, sm);
  if (threadIdx.x == 0) {
    const double nd = sqrt(s);
    res[AA_R_NORMD] = nd;
    // NaN-safe: reject unless the new residual is provably no larger
    *bad = (nd <= factor * res[AA_R_NORMG]) ? 0 : 1;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_fin_safeguard(const double *part, int np, double factor,
                                                               double *res, int *bad) {
  d_fin_safeguard(part, np, factor, res, bad);
}
// roll back to the pre-AA iterate when the safeguard fired
__device__ __forceinline__ void d_aa_restore(double *f_new, double *x_new, const double *__restrict__ af,
                                             const double *__restrict__ ax, long dim, const int *bad) {
  if (!*bad) return;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    f_new[i] = af[i];
    x_new[i] = ax[i];
  }
}
__global__ __launch_bounds__(kVecThreads) void k_aa_restore(double *f_new, double *x_new,
                                                            const double *__restrict__ af,
                                                            const double *__restrict__ ax, long dim,
                                                            const int *bad) {
  d_aa_restore(f_new, x_new, af, ax, dim, bad);
}

// ================================================================================================ host object
// Mirrors the control flow of aa.c: apply() / safeguard() / reset(); all vectors are device pointers on `stream`.
struct DeviceAa {
  long dim = 0;
  int mem = 0, type1 = 1, iter = 0, success = 0;
  double relaxation = 1.0, regularization = 1e-8, safeguard_factor = 1.0, max_weight_norm = 1e10;
  bool tsqr = true;
  hipStream_t stream = nullptr;
  DevBuf<double> x, f, gprev, S, Y, D, npart, spart, part, out, res, gamma, rbuf[2];
  DevBuf<int> bad;
  double *h_pin = nullptr;  // pinned: 1 + 3 * kAaMaxMem (Gram partial results) / AA_R_COUNT (TSQR verdict)
  bool owns_pin = true;     // false: h_pin points into the owning workspace's pinned block
  std::vector<double> M;    // Gram path: raw mem x mem system matrix (col-major), maintained incrementally
  ScsAaStats st{};
  bool pending_safeguard = false;
  double last_norm_g = 0;
  std::vector<double> last_gamma;  // weights of the last solve (tests: scs_hip_aa_last_gamma)

  DeviceAa() = default;
  DeviceAa(const DeviceAa &) = delete;
  DeviceAa &operator=(const DeviceAa &) = delete;
  ~DeviceAa() { if (h_pin && owns_pin) (void)hipHostFree(h_pin); }

  static bool tsqr_default() { return !opts().aa_gram; }  // SCS_HIP_AA=gram: incremental Gram update + host solve (A/B, tests)
  int nbl() const { return vec_blocks(dim); }

  void init(long dim_, int mem_, int type1_, double regularization_, double relaxation_, double safeguard_factor_,
            double max_weight_norm_, hipStream_t s) {
    dim = dim_; mem = mem_; type1 = type1_ ? 1 : 0; regularization = regularization_; relaxation = relaxation_;
    safeguard_factor = safeguard_factor_; max_weight_norm = max_weight_norm_; stream = s;
    iter = 0; success = 0; st = ScsAaStats{}; pending_safeguard = false;
    tsqr = tsqr_default() && mem <= kAaMaxMem;  // (longer histories: the Gram path, a block of kAaMaxMem columns per launch)
    if (mem <= 0) return;
    for (DevBuf<double> *b : {&x, &f, &gprev}) b->alloc_zero((size_t)dim, s);
    for (DevBuf<double> *b : {&S, &Y, &D}) b->alloc_zero((size_t)dim * mem, s);
    npart.alloc_zero(kMaxVecBlocks, s);
    spart.alloc_zero(kMaxVecBlocks, s);
    res.alloc_zero(AA_R_COUNT, s);
    bad.alloc_zero(1, s);
    if (mem > kAaMaxMem) {  // the Gram record (1 + 3 mem doubles) outgrows the shared pinned block
      if (h_pin && owns_pin) (void)hipHostFree(h_pin);
      h_pin = nullptr;
      owns_pin = true;
    }
    if (!h_pin) HIP_CHECK(hipHostMalloc((void **)&h_pin, sizeof(double) * std::max(256, 2 + 3 * mem)));
    if (tsqr) {
      const int c = ncols();
      const size_t cap = (size_t)c * kTsqrWavesFast * mem;
      rbuf[0].alloc_zero(cap, s);
      rbuf[1].alloc_zero((size_t)c * 64 * mem, s);  // (second level: at most 64 chains)
    } else {
      part.alloc_zero((size_t)3 * mem * kMaxVecBlocks, s);
      out.alloc_zero(1 + 3 * (size_t)mem, s);
      gamma.alloc_zero((size_t)mem, s);
      M.assign((size_t)mem * mem, 0.0);
    }
  }
  void reset() { iter = 0; }
  int ncols() const { return type1 ? 2 * mem + 1 : mem + 1; }

  static constexpr int kTsqrWaves1 = 1024, kTsqrWaves2 = 16;
  // chains of the register-tile level-1 kernel: enough wavefronts to put 2-3 on every SIMD of the chip (SCS_HIP_AA_WAVES1: lab)
  static constexpr int kTsqrWavesFast = 4096;
  static int fast_waves() {
    const int t = opts().aa_waves1;  // (labs knob)
    return t > 0 && t <= kTsqrWavesFast ? t : 2048;
  }

  // One launch of the TSQR reduction: the tall matrix it reads, its geometry and where its stacked triangles go
  struct TsqrLevel {
    AaTall W;
    int rho;
    long tiles_per_wave, nw, out_ld;
    double *out;
    size_t lds;
    int fast = 0;  // aa_tsqr_fast_kind: level 1 of the default histories runs the register-tile kernel (1: c = 21, 2: c = 11)
  };
  // [L | Y | g] -> final npiv x c triangle (column-major, ld = npiv): the launches of the fixed reduction tree, in
  // order; the last level's `out` holds the triangle.  (Also read by the grouped solve, batch.hpp.)
  std::vector<TsqrLevel> tsqr_levels(int len) const {
    std::vector<TsqrLevel> lv;
    AaTall W{};
    W.L = type1 ? S.p : Y.p; W.Y = type1 ? Y.p : nullptr; W.g = gprev.p;
    W.ld = dim; W.rows = dim; W.nL = len; W.nY = type1 ? len : 0; W.c = W.nL + W.nY + 1; W.npiv = len;
    const int c = W.c, rho = aa_pick_rho(c);
    const size_t lds = aa_tsqr_lds(c, len, rho);
    const bool fast_on = opts().aa_fast;  // (labs switch: the LDS kernel of round 2)
    int level = 0, dst = 0;
    while (true) {
      const int fast = fast_on ? aa_tsqr_fast_kind(c, len) : 0;  // (every level: the stacked triangles have the same c columns and pivots)
      const int TR = 64 * (fast ? kAaFastRho : rho);
      const long ntiles = std::max(1L, (W.rows + TR - 1) / TR);
      const long cap = fast ? (level == 0 ? fast_waves() : 64) : (level == 0 ? kTsqrWaves1 : kTsqrWaves2);
      long nw = ntiles <= 8 ? 1 : std::min(ntiles, cap);
      const long tpw = (ntiles + nw - 1) / nw;
      nw = (ntiles + tpw - 1) / tpw;
      double *o = rbuf[dst].p;
      const long out_ld = nw * len;
      lv.push_back(TsqrLevel{W, rho, tpw, nw, out_ld, o, lds, fast});
      if (nw == 1) return lv;
      W.L = o; W.Y = nullptr; W.g = nullptr; W.ld = out_ld; W.rows = out_ld; W.nL = c; W.nY = 0;
      dst ^= 1;
      ++level;
    }
  }
  const double *tsqr_factor(int len) {
    const std::vector<TsqrLevel> lv = tsqr_levels(len);
    for (const TsqrLevel &L : lv) {
      const dim3 gf((unsigned)((L.nw + kAaFastWaves - 1) / kAaFastWaves)), bf(64 * kAaFastWaves);
      if (L.fast == 1) hipLaunchKernelGGL((k_aa_tsqr_fast<21, 10>), gf, bf, 0, stream, L.W, L.tiles_per_wave, L.nw, L.out, L.out_ld);
      else if (L.fast == 2) hipLaunchKernelGGL((k_aa_tsqr_fast<11, 10>), gf, bf, 0, stream, L.W, L.tiles_per_wave, L.nw, L.out, L.out_ld);
      else hipLaunchKernelGGL(k_aa_tsqr, dim3((unsigned)L.nw), dim3(64), L.lds, stream, L.W, L.rho, L.tiles_per_wave, L.out, L.out_ld);
    }
    return lv.back().out;
  }

  // What one call has to do, from the control state alone (aa.c's apply): 0 nothing (no memory), 1 seed the history,
  // 2 extend it (still filling), 3 extend + solve + extrapolate.  len / idx: columns in use / column written.
  // The caller runs the kernels, hands the solve's record to complete() (mode 3) and then advances `iter`.
  int plan(int &len, int &idx) {
    success = 0;
    len = idx = 0;
    if (mem <= 0) return 0;
    st.iter++;
    if (iter == 0) return 1;
    len = std::min(iter, mem);
    idx = (iter - 1) % mem;
    return iter >= mem ? 3 : 2;
  }
  // res = the AA_R_* record k_aa_solve left (read back by the caller); returns aa_norm
  double complete(const double *res, int len) {
    const double aa_norm = verdict((int)res[AA_R_RANK], (int)res[AA_R_CODE], res[AA_R_NORM], res[AA_R_REG]);
    last_gamma.assign(res + AA_R_GAMMA, res + AA_R_GAMMA + len);
    last_norm_g = res[AA_R_NORMG];
    return aa_norm;
  }

  // f = current map output F(x) (device, may be overwritten with the extrapolated iterate), x = map input.
  // Returns aa_norm with aa.c's sign convention (0: nothing done, < 0: rejected).
  double apply(double *fdev, const double *xdev) {
    double aa_norm = 0;
    int len, idx;
    const int mode = plan(len, idx);
    if (mode == 0) return aa_norm;
    const int nb = nbl();
    if (mode == 1) {
      hipLaunchKernelGGL(k_aa_seed, dim3(nb), dim3(kVecThreads), 0, stream, xdev, fdev, x.p, f.p, gprev.p, dim);
      iter++;
      return aa_norm;
    }
    hipLaunchKernelGGL(k_aa_update, dim3(nb), dim3(kVecThreads), 0, stream, xdev, fdev, x.p, f.p, gprev.p, S.p, Y.p, D.p, dim,
                       idx, npart.p);
    if (tsqr) {
      if (mode == 3) {
        const double *R = tsqr_factor(len);
        hipLaunchKernelGGL(k_aa_solve, dim3(1), dim3(64), 0, stream, R, len, ncols(), type1, regularization, max_weight_norm,
                           (const double *)npart.p, nb, res.p);
        hipLaunchKernelGGL(k_aa_apply, dim3(nb), dim3(kVecThreads), 0, stream, fdev, D.p, S.p, x.p, res.p + AA_R_GAMMA, dim, len,
                           relaxation, (const double *)res.p + AA_R_OK);
        HIP_CHECK(hipMemcpyAsync(h_pin, res.p, sizeof(double) * AA_R_COUNT, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        aa_norm = complete(h_pin, len);
      }
    } else {
      const double *L = type1 ? S.p : Y.p;
      for (int j0 = 0; j0 < len; j0 += kAaMaxMem)
        hipLaunchKernelGGL(k_aa_dots, dim3(nb), dim3(kVecThreads), 0, stream, L, Y.p, gprev.p, dim, len, idx, part.p, j0, mem);
      hipLaunchKernelGGL(k_fin_aa, dim3(1), dim3(kVecThreads), 0, stream, npart.p, nb, part.p, nb, len, out.p, res.p, mem);
      HIP_CHECK(hipMemcpyAsync(h_pin, out.p, sizeof(double) * (1 + 3 * (size_t)mem), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      const double *o = h_pin;
      last_norm_g = std::sqrt(o[0]);
      for (int j = 0; j < len; ++j) {
        M[idx + mem * j] = o[1 + 0 * mem + j];  // row idx
        M[j + mem * idx] = o[1 + 1 * mem + j];  // col idx
      }
      if (iter >= mem) {
        std::vector<double> A((size_t)len * len), w(len);
        double nrm = 0.;
        for (int j = 0; j < len; ++j)
          for (int i = 0; i < len; ++i) {
            A[i + (size_t)len * j] = M[i + mem * j];
            nrm += A[i + (size_t)len * j] * A[i + (size_t)len * j];
          }
        const double reg = regularization * std::sqrt(nrm);
        if (regularization > 0)
          for (int i = 0; i < len; ++i) A[i + (size_t)len * i] += reg;
        for (int j = 0; j < len; ++j) w[j] = o[1 + 2 * mem + j];
        const int rank = dense_solve(A.data(), w.data(), len);
        double nw = 0.;
        int code = AA_CODE_OK;
        if (rank == 0) code = AA_CODE_RANK0;
        else if (rank < len) code = AA_CODE_LAPACK;
        else {
          for (int j = 0; j < len; ++j) nw += w[j] * w[j];
          nw = std::sqrt(nw);
          if (!std::isfinite(nw)) code = AA_CODE_NONFINITE;
          else if (nw >= max_weight_norm) code = AA_CODE_WEIGHT;
        }
        aa_norm = verdict(rank, code, nw, reg);
        last_gamma = w;
        if (code == AA_CODE_OK) {
          HIP_CHECK(hipMemcpyAsync(gamma.p, w.data(), sizeof(double) * len, hipMemcpyHostToDevice, stream));
          hipLaunchKernelGGL(k_aa_apply, dim3(nb), dim3(kVecThreads), 0, stream, fdev, D.p, S.p, x.p, gamma.p, dim, len, relaxation,
                             (const double *)nullptr);
          HIP_CHECK(hipStreamSynchronize(stream));  // w is a local
        }
      }
    }
    iter++;
    return aa_norm;
  }

  // statistics and control state after a solve (same branches as aa.c's solve)
  double verdict(int rank, int code, double nw, double reg) {
    st.last_regularization = reg;
    st.last_rank = rank;
    switch (code) {
      case AA_CODE_RANK0: st.n_reject_rank0++; success = 0; iter = 0; return -1.;
      case AA_CODE_LAPACK: st.n_reject_lapack++; success = 0; iter = 0; return -1.;
      default: break;
    }
    st.last_aa_norm = nw;
    if (code == AA_CODE_NONFINITE) { st.n_reject_nonfinite++; success = 0; iter = 0; return -1.; }
    if (code == AA_CODE_WEIGHT) { st.n_reject_weight_cap++; success = 0; iter = 0; return -nw; }
    success = 1;
    st.n_accept++;
    return nw;
  }  // (as in aa.c the reset happens inside solve and apply's iter++ follows: the rejected call's (x, f) is the new seed)

  // f_new = F(x_new) after an accepted extrapolation x_new.  Enqueues the residual test and the conditional roll-back;
  // the verdict lands in *bad_dev (device int) — hand it to safeguard_verdict() once read.  false: nothing to test.
  bool safeguard(double *f_new, double *x_new, int *bad_dev) {
    if (!success) return false;
    success = 0;
    const int nb = nbl();
    double *p = spart.p;
    hipLaunchKernelGGL(k_aa_diffsq, dim3(nb), dim3(kVecThreads), 0, stream, (const double *)x_new, (const double *)f_new, dim, p);
    hipLaunchKernelGGL(k_fin_safeguard, dim3(1), dim3(kVecThreads), 0, stream, (const double *)p, nb, safeguard_factor, res.p, bad_dev);
    hipLaunchKernelGGL(k_aa_restore, dim3(nb), dim3(kVecThreads), 0, stream, f_new, x_new, (const double *)f.p, (const double *)x.p,
                       dim, (const int *)bad_dev);
    pending_safeguard = true;
    return true;
  }
  void safeguard_verdict(bool rejected) {
    pending_safeguard = false;
    if (rejected) { st.n_safeguard_reject++; reset(); }
  }

  static int dense_solve(double *A, double *rhs, int nn) {
    int rank = 0;
    for (int k = 0; k < nn; ++k) {
      int piv = k;
      double mx = std::fabs(A[k + nn * k]);
      for (int i = k + 1; i < nn; ++i) {
        const double a = std::fabs(A[i + nn * k]);
        if (a > mx) { mx = a; piv = i; }
      }
      if (!(mx > 0.) || !std::isfinite(mx)) return rank;
      rank++;
      if (piv != k) {
        for (int j = 0; j < nn; ++j) std::swap(A[k + nn * j], A[piv + nn * j]);
        std::swap(rhs[k], rhs[piv]);
      }
      for (int i = k + 1; i < nn; ++i) {
        const double f = A[i + nn * k] / A[k + nn * k];
        if (f == 0.) continue;
        for (int j = k + 1; j < nn; ++j) A[i + nn * j] -= f * A[k + nn * j];
        rhs[i] -= f * rhs[k];
      }
    }
    for (int k = nn - 1; k >= 0; --k) {
      double s = rhs[k];
      for (int j = k + 1; j < nn; ++j) s -= A[k + nn * j] * rhs[j];
      rhs[k] = s / A[k + nn * k];
    }
    return rank;
  }
};

}  // namespace scship
