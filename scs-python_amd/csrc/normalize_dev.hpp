// normalize_dev.hpp — K12: data equilibration on the device (init time).
//
// Plays the role of scs_source/src/normalize.c + the normalisation helpers of
// scs_source/linsys/scs_matrix.c (R:meson.build:192,200; absent).  `normalize` defaults to true
// (R:scs/scsobject.h:797).  Algorithm: 25 Ruiz (inf-norm) passes + 1 l2 pass on
// [P A'; A 0]; row scalings made constant inside every non-separable cone block (max / mean);
// factors clamped to [1e-4, 1e4];
//     A_hat = D A E,  P_hat = E P E,  b_hat = sigma D b,  c_hat = sigma E c.
//
// All three resident layouts are scaled in place: CSR(A) gives the row norms, CSR(A') (= the caller's
// CSC) the column norms, the full symmetric CSR(P) its contribution to E.  One pass = three
// segmented reductions + three rescale sweeps, each one streaming pass over a matrix (HBM-bound,
// ~0.1 ms at nnz = 2e7), 26 passes.  Row sums run in CSR order = ascending column, i.e. the same
// order as the oracle's CSC loops, so D and E agree with the CPU restatement to the last bit
// (except the l2-pass contribution of P and the mean over large cone blocks: ~1 ulp).
#pragma once
#include "common.hpp"
#include "spmv.hpp"
#include "vec.hpp"

namespace scship {

// out[r] = max |val| (l2 = 0) or sqrt(sum val^2) (l2 = 1) over CSR row r; thread per row
__global__ __launch_bounds__(kVecThreads) void k_row_norm(const int *__restrict__ rowptr, const double *__restrict__ val,
                                                          int rows, int l2, double *out) {
  for (long r = (long)blockIdx.x * kVecThreads + threadIdx.x; r < rows; r += (long)gridDim.x * kVecThreads) {
    double acc = 0.;
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      const double v = fabs(val[p]);
      acc = l2 ? __dadd_rn(acc, __dmul_rn(v, v)) : fmax(acc, v);  // no FMA contraction: same bits as the CPU restatement
    }
    out[r] = acc;
  }
}

// One equilibration sweep over a CSR matrix, coalesced (row blocks of spmv.hpp: lanes stream the nonzeros, the
// row phases work on LDS):  val[p] <- val[p] * (rs[row] * cs[col[p]])  (skipped when rs == nullptr), then
// out[r] = max |val| (l2 = 0) or sum val^2 (l2 = 1) over the UPDATED row r (skipped when l2 < 0).
// Fusing the norm of the next pass into the rescale of this one halves the passes over the matrices; per-row
// order is ascending p as in k_row_norm / k_rescale, so D and E keep their bits (rows longer than a block:
// workgroup reduction, ~1 ulp in the l2 pass).
__device__ __forceinline__ void d_rescale_norm(const CsrView &A, double *val, const double *__restrict__ rs, const double *__restrict__ cs, int l2,
                                               double *out, int blk, double *buf, double *red) {
  const int tid = threadIdx.x;
  const int4 bi = A.blk[blk];
  const int r0 = bi.x, r1 = bi.y, p0 = bi.z, p1 = bi.w, nnz = p1 - p0;
  if (nnz <= kNnzPerWg) {
    int ra[kRowsPerLane], re[kRowsPerLane];
#pragma unroll
    for (int j = 0; j < kRowsPerLane; ++j) {
      const int r = r0 + tid + j * kSpmvThreads;
      ra[j] = r < r1 ? A.rowptr[r] - p0 : 0;
      re[j] = r < r1 ? A.rowptr[r + 1] - p0 : 0;
      if (rs && r < r1) {
        const double f = rs[r];
        for (int k = ra[j]; k < re[j]; ++k) buf[k] = f;  // row factor of every nonzero of the row
      }
    }
    __syncthreads();
    for (int k = tid; k < nnz; k += kSpmvThreads) {
      double v = val[p0 + k];
      if (rs) {
        v = __dmul_rn(v, __dmul_rn(buf[k], cs[A.col[p0 + k]]));
        val[p0 + k] = v;
      }
      v = fabs(v);
      buf[k] = l2 > 0 ? __dmul_rn(v, v) : v;
    }
    if (l2 < 0) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRowsPerLane; ++j) {
      const int r = r0 + tid + j * kSpmvThreads;
      if (r < r1) {
        double acc = 0.;
        for (int k = ra[j]; k < re[j]; ++k) acc = l2 > 0 ? __dadd_rn(acc, buf[k]) : fmax(acc, buf[k]);
        out[r] = acc;
      }
    }
  } else {  // one long row
    const double f = rs ? rs[r0] : 1.0;
    double acc = 0.;
    for (int p = p0 + tid; p < p1; p += kSpmvThreads) {
      double v = val[p];
      if (rs) {
        v = __dmul_rn(v, __dmul_rn(f, cs[A.col[p]]));
        val[p] = v;
      }
      v = fabs(v);
      acc = l2 > 0 ? __dadd_rn(acc, __dmul_rn(v, v)) : fmax(acc, v);
    }
    if (l2 < 0) return;
    acc = l2 > 0 ? block_sum<kSpmvThreads>(acc, red) : block_max<kSpmvThreads>(acc, red);
    if (tid == 0) out[r0] = acc;
  }
}
__global__ __launch_bounds__(kSpmvThreads) void k_rescale_norm(CsrView A, double *val, const double *__restrict__ rs,
                                                               const double *__restrict__ cs, int l2, double *out) {
  __shared__ double buf[kNnzPerWg];
  __shared__ double red[kSpmvThreads / 64];
  d_rescale_norm(A, val, rs, cs, l2, out, (int)blockIdx.x, buf, red);
}
// the sweeps of a pass over A (rows -> next D norms) and A' (columns -> next E norms) [and P] do not depend on each other: one launch,
// the workgroups [0, n1) on the first matrix, [n1, n1 + n2) on the second, the rest on the third (round 5, late: a small problem's
// scs_init is a chain of dependent dispatches)
__global__ __launch_bounds__(kSpmvThreads) void k_rescale_norm3(CsrView A1, double *v1, const double *__restrict__ rs1, const double *__restrict__ cs1,
                                                                double *o1, int n1, CsrView A2, double *v2, const double *__restrict__ rs2,
                                                                const double *__restrict__ cs2, double *o2, int n2, CsrView A3, double *v3,
                                                                const double *__restrict__ rs3, const double *__restrict__ cs3, double *o3, int l2) {
  __shared__ double buf[kNnzPerWg];
  __shared__ double red[kSpmvThreads / 64];
  const int b = (int)blockIdx.x;
  if (b < n1) d_rescale_norm(A1, v1, rs1, cs1, l2, o1, b, buf, red);
  else if (b < n1 + n2) d_rescale_norm(A2, v2, rs2, cs2, l2, o2, b - n1, buf, red);
  else d_rescale_norm(A3, v3, rs3, cs3, l2, o3, b - n1 - n2, buf, red);
}

// Et = combine(Et_A, Et_P): max for the inf passes, sum of squares for the l2 pass (before the sqrt)
__global__ __launch_bounds__(kVecThreads) void k_combine(double *a, const double *__restrict__ b, int n, int l2) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads)
    a[i] = l2 ? a[i] + b[i] : fmax(a[i], b[i]);
}
__global__ __launch_bounds__(kVecThreads) void k_sqrt_inplace(double *a, int n) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) a[i] = sqrt(a[i]);
}

// make Dt constant over each non-separable cone block: max (inf passes) or mean (l2 pass); one wave per block,
// blocks up to 64 rows are summed sequentially by lane 0 (the oracle's order), larger ones by a fixed tree
__global__ __launch_bounds__(kVecThreads) void k_enforce_blocks(double *Dt, const int *__restrict__ boff,
                                                                const int *__restrict__ blen, int nblocks, int use_mean) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (kVecThreads / 64) + (threadIdx.x >> 6);
  if (b >= nblocks) return;
  const int len = blen[b];
  if (len <= 0) return;
  double *d = Dt + boff[b];
  double w = 0.;
  if (len <= 64) {
    if (lane == 0) {
      for (int j = 0; j < len; ++j) w = use_mean ? w + d[j] : fmax(w, fabs(d[j]));
      if (use_mean) w /= (double)len;
    }
    w = __shfl(w, 0, 64);
  } else {
    for (int j = lane; j < len; j += 64) w = use_mean ? w + d[j] : fmax(w, fabs(d[j]));
    w = use_mean ? wave_sum(w) : wave_max(w);
    w = __shfl(w, 0, 64);
    if (use_mean) w /= (double)len;
  }
  for (int j = lane; j < len; j += 64) d[j] = w;
}

// t = 1 / sqrt(limit(t)); acc *= t
__global__ __launch_bounds__(kVecThreads) void k_invsqrt_acc(double *t, double *acc, int n) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) {
    double x = t[i];
    x = x < 1e-4 ? 1.0 : x;
    x = x > 1e4 ? 1e4 : x;
    const double s = sqrt(x);
    const double f = s < 1e-18 ? 1.0 / 1e-18 : 1.0 / s;
    t[i] = f;
    acc[i] *= f;
  }
}

// One launch for what follows the norms of a pass (round 5, late: scs_init of a small problem is bound by the NUMBER of runtime calls — 134 launches
// of an equilibration, 16 setup threads of a batch contending for the runtime): [the square roots of the l2 pass,] the block rule of
// k_enforce_blocks and both k_invsqrt_acc sweeps.  Workgroups [0, nbD): rows in front of the first cone block, element by element;
// [nbD, nbD + nbE): the columns; the rest: one wavefront per cone block (EVERY block behind the separable rows, the one-row ones too — for
// those the block rule is the identity), which reduces the block exactly like k_enforce_blocks and finishes its rows itself.
// Same operations on the same values in the same order: D and E keep their bits.
__device__ __forceinline__ double d_invsqrt_limit(double x) {
  x = x < 1e-4 ? 1.0 : x;
  x = x > 1e4 ? 1e4 : x;
  const double s = sqrt(x);
  return s < 1e-18 ? 1.0 / 1e-18 : 1.0 / s;
}
__global__ __launch_bounds__(kVecThreads) void k_pass_finish(double *Dt, double *D, int prefix, double *Et, double *E, int n, const int *__restrict__ boff,
                                                             const int *__restrict__ blen, int nblocks, int l2, int nbD, int nbE) {
  const int b = blockIdx.x;
  if (b < nbD + nbE) {
    const bool rows = b < nbD;
    double *t = rows ? Dt : Et, *acc = rows ? D : E;
    const int cnt = rows ? prefix : n, nb = rows ? nbD : nbE, bb = rows ? b : b - nbD;
    for (long i = (long)bb * kVecThreads + threadIdx.x; i < cnt; i += (long)nb * kVecThreads) {
      double x = t[i];
      if (l2) x = sqrt(x);
      const double f = d_invsqrt_limit(x);
      t[i] = f;
      acc[i] *= f;
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int blk = (b - nbD - nbE) * (kVecThreads / 64) + (threadIdx.x >> 6);
  if (blk >= nblocks) return;
  const int len = blen[blk];
  if (len <= 0) return;
  double *d = Dt + boff[blk], *a = D + boff[blk];
  double w = 0.;
  if (len <= 64) {
    // every lane fetches one entry, then the entries are added in index order (the oracle's order, as lane 0's loop over memory did in
    // k_enforce_blocks — 50 dependent round trips for a 50-row cone, 15 us per launch: the longest link in the chain of a small scs_init)
    double v = lane < len ? d[lane] : 0.;
    if (l2) v = sqrt(v);
    for (int j = 0; j < len; ++j) {
      const double vj = __shfl(v, j, 64);
      w = l2 ? w + vj : fmax(w, fabs(vj));
    }
    if (l2) w /= (double)len;
  } else {
    for (int j = lane; j < len; j += 64) {
      const double v = l2 ? sqrt(d[j]) : d[j];
      w = l2 ? w + v : fmax(w, fabs(v));
    }
    w = l2 ? wave_sum(w) : wave_max(w);
    w = __shfl(w, 0, 64);
    if (l2) w /= (double)len;
  }
  const double f = d_invsqrt_limit(w);
  for (int j = lane; j < len; j += 64) {
    d[j] = f;
    a[j] *= f;
  }
}

// val[p] *= rs[row] * cs[col[p]]
__global__ __launch_bounds__(kVecThreads) void k_rescale(const int *__restrict__ rowptr, const int *__restrict__ col, double *val,
                                                         int rows, const double *__restrict__ rs, const double *__restrict__ cs) {
  for (long r = (long)blockIdx.x * kVecThreads + threadIdx.x; r < rows; r += (long)gridDim.x * kVecThreads) {
    const double f = rs[r];
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) val[p] = __dmul_rn(val[p], __dmul_rn(f, cs[col[p]]));
  }
}

// diag[r] = sum of the entries (r, r) of a CSR matrix
__global__ __launch_bounds__(kVecThreads) void k_csr_diag(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                          const double *__restrict__ val, int rows, double *diag) {
  for (long r = (long)blockIdx.x * kVecThreads + threadIdx.x; r < rows; r += (long)gridDim.x * kVecThreads) {
    double d = 0.;
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p)
      if (col[p] == r) d += val[p];
    diag[r] = d;
  }
}

// v[i] *= s[i]; partial max |v|
__global__ __launch_bounds__(kVecThreads) void k_scale_by_vec(double *v, const double *__restrict__ s, int n, double *part) {
  __shared__ double sm[kVecThreads / 64];
  double mx = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) {
    const double x = v[i] * s[i];
    v[i] = x;
    mx = fmax(mx, fabs(x));
  }
  mx = block_max<kVecThreads>(mx, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = mx;
}
__global__ __launch_bounds__(kVecThreads) void k_scale_scalar(double *v, double a, long n) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) v[i] *= a;
}
__global__ __launch_bounds__(kVecThreads) void k_fill(double *v, double a, long n) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) v[i] = a;
}
// dst[k] = src[perm[k]] (perm < 0: zero padding) — refreshes the slab copy from the scaled CSR values
__global__ __launch_bounds__(kVecThreads) void k_gather_vals(double *dst, const double *__restrict__ src, const int *__restrict__ perm,
                                                             long n) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) {
    const int p = perm[i];
    dst[i] = p >= 0 ? src[p] : 0.0;
  }
}

}  // namespace scship
