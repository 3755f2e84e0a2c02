// aa.hpp — K10: Anderson acceleration on the device-resident iterate v (length l = n+m+1).
//
// Plays the role of scs_source/src/aa.c (named at R:meson.build:187; absent); knobs
// R:README.md:98-104, statistics R:scs/scsobject.h:1096-1107.
//
// Data layout: S, Y, D are tall-skinny l x mem column-major matrices in HBM
// (3 * l * mem * 8 B; 720 MB at l = 3e6, mem = 10 — trivial against 288 GB).
// Per call exactly ONE column changes, so the mem x mem system matrix
// M = S'Y (type-I) / Y'Y (type-II) is updated incrementally: one fused pass
// streams S and Y once (coalesced, HBM-bound) and produces the new row, the new
// column and S'g with fixed-order two-stage reductions.  The tiny dense solve
// (mem <= 32) is done by the host from 3*mem reduced scalars — the only values
// that cross PCIe — and the extrapolation f -= D gamma is one more streaming pass.
#pragma once
#include "common.hpp"
#include "vec.hpp"

namespace scship {

constexpr int kAaMaxMem = 32;

// first call after a reset: x_prev = x, f_prev = f, g_prev = x - f
__global__ __launch_bounds__(kVecThreads) void k_aa_seed(const double *__restrict__ x, const double *__restrict__ f, double *ax,
                                                         double *af, double *gprev, long dim) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    const double xi = x[i], fi = f[i];
    ax[i] = xi;
    af[i] = fi;
    gprev[i] = xi - fi;
  }
}

// g = x - f; s = x - x_prev; d = f - f_prev; y = g - g_prev; store column idx of S,D,Y;
// roll x_prev,f_prev,g_prev; partial ||g||^2
__global__ __launch_bounds__(kVecThreads) void k_aa_update(const double *__restrict__ x, const double *__restrict__ f, double *ax,
                                                           double *af, double *gprev, double *S, double *Y, double *D, long dim,
                                                           int idx, double *part) {
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

// one streaming pass over L (= S or Y) and Y:  row[j] = L_idx . Y_j,  col[j] = L_j . Y_idx,  w[j] = L_j . g
// partial layout: part[(k*kAaMaxMem + j) * nb + b], k = 0 (row), 1 (col), 2 (w)
__global__ __launch_bounds__(kVecThreads) void k_aa_dots(const double *__restrict__ L, const double *__restrict__ Y,
                                                         const double *__restrict__ g, long dim, int len, int idx, double *part) {
  __shared__ double sm[kVecThreads / 64];
  double row[kAaMaxMem], col[kAaMaxMem], w[kAaMaxMem];
#pragma unroll
  for (int j = 0; j < kAaMaxMem; ++j) row[j] = col[j] = w[j] = 0.;
  const double *Li = L + (size_t)dim * idx, *Yi = Y + (size_t)dim * idx;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    const double li = Li[i], yi = Yi[i], gi = g[i];
#pragma unroll
    for (int j = 0; j < kAaMaxMem; ++j) {
      if (j < len) {
        const double Lj = L[(size_t)dim * j + i], Yj = Y[(size_t)dim * j + i];
        row[j] += li * Yj;
        col[j] += Lj * yi;
        w[j] += Lj * gi;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kAaMaxMem; ++j) {
    if (j < len) {  // len is uniform across the grid
      const double a = block_sum<kVecThreads>(row[j], sm);
      const double b = block_sum<kVecThreads>(col[j], sm);
      const double c = block_sum<kVecThreads>(w[j], sm);
      if (threadIdx.x == 0) {
        part[((size_t)(0 * kAaMaxMem + j)) * gridDim.x + blockIdx.x] = a;
        part[((size_t)(1 * kAaMaxMem + j)) * gridDim.x + blockIdx.x] = b;
        part[((size_t)(2 * kAaMaxMem + j)) * gridDim.x + blockIdx.x] = c;
      }
    }
  }
}

// out[0] = sum of norm partials (||g||^2); out[1 + k*kAaMaxMem + j] = reduced dots
__global__ __launch_bounds__(kVecThreads) void k_fin_aa(const double *npart, int nnp, const double *part, int np, int len,
                                                        double *out, double *sc) {
  __shared__ double sm[kVecThreads / 64];
  const double ng = part_sum(npart, nnp, sm);
  if (threadIdx.x == 0) {
    out[0] = ng;
    sc[S_AA_NORMG] = sqrt(ng);
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k)
    for (int j = 0; j < len; ++j) {
      const double v = part_sum(part + ((size_t)(k * kAaMaxMem + j)) * np, np, sm);
      if (threadIdx.x == 0) out[1 + k * kAaMaxMem + j] = v;
      __syncthreads();
    }
}

// f -= D gamma;  optional relaxation: f = beta f + (1-beta) (x - S gamma)
__global__ __launch_bounds__(kVecThreads) void k_aa_apply(double *f, const double *__restrict__ D, const double *__restrict__ S,
                                                          const double *__restrict__ xcur, const double *__restrict__ gamma,
                                                          long dim, int len, double relaxation) {
  double gm[kAaMaxMem];
#pragma unroll
  for (int j = 0; j < kAaMaxMem; ++j) gm[j] = j < len ? gamma[j] : 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    double fi = f[i], xs = 0.;
#pragma unroll
    for (int j = 0; j < kAaMaxMem; ++j)
      if (j < len) fi -= gm[j] * D[(size_t)dim * j + i];
    if (relaxation != 1.0) {
#pragma unroll
      for (int j = 0; j < kAaMaxMem; ++j)
        if (j < len) xs += gm[j] * S[(size_t)dim * j + i];
      fi = relaxation * fi + (1. - relaxation) * (xcur[i] - xs);
    }
    f[i] = fi;
  }
}

// safeguard: partial ||x_new - f_new||^2
__global__ __launch_bounds__(kVecThreads) void k_aa_diffsq(const double *__restrict__ a, const double *__restrict__ b, long dim,
                                                           double *part) {
  __shared__ double sm[kVecThreads / 64];
  double acc = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    const double d = a[i] - b[i];
    acc += d * d;
  }
  acc = block_sum<kVecThreads>(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(kVecThreads) void k_fin_safeguard(const double *part, int np, double factor, double *sc, int *fl) {
  __shared__ double sm[kVecThreads / 64];
  const double s = part_sum(part, np, sm);
  if (threadIdx.x == 0) {
    const double nd = sqrt(s);
    sc[S_AA_NORMD] = nd;
    // NaN-safe: reject unless the new residual is provably no larger
    fl[F_SAFE_BAD] = (nd <= factor * sc[S_AA_NORMG]) ? 0 : 1;
  }
}
// roll back to the pre-AA iterate when the safeguard fired
__global__ __launch_bounds__(kVecThreads) void k_aa_restore(double *f_new, double *x_new, const double *__restrict__ af,
                                                            const double *__restrict__ ax, long dim, const int *fl) {
  if (!fl[F_SAFE_BAD]) return;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < dim; i += (long)gridDim.x * kVecThreads) {
    f_new[i] = af[i];
    x_new[i] = ax[i];
  }
}

}  // namespace scship
