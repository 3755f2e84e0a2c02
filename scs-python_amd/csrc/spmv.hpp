// spmv.hpp — K1/K2/K3: the A, A' and P mat-vecs of the indirect KKT solve.
//
// Plays the role of scs_source/linsys/scs_matrix.c (accum_by_a / accum_by_atrans /
// accum_by_p; named at R:meson.build:199-202, source absent) and replaces the
// cuSPARSE calls of the reference's GPU_INDIRECT backend (R:legacy_setup.py:263).
//
// Layout in HBM: the caller's CSC(A) is used zero-conversion as
// CSR(A') for x-space outputs; an explicit CSR(A) is built once at init for
// y-space outputs; P is expanded to a full symmetric CSR.  fp64 values, int32
// indices (R:meson.build:172-174).
//
// Kernel: CSR-stream.  A workgroup owns a contiguous run of rows whose nonzeros
// fit in LDS (kNnzPerWg).  All 256 lanes stream val/col with unit stride
// (coalesced HBM reads), multiply by the gathered x (L2/MALL resident) and stage
// the products in LDS; then each lane sums the LDS segment of one row in CSR
// order.  Summation order per row is fixed => bit-deterministic, and equal to
// the column-ordered CPU scatter/gather of the oracle.
// Algorithmic bytes per launch: 12*nnz + 4*(rows+1) + 8*cols + 8*rows.
#pragma once
#include "common.hpp"

namespace scship {

constexpr int kSpmvThreads = 256;
constexpr int kNnzPerWg = 2048;  // 16 KiB of LDS products per workgroup

// Device view of a CSR matrix plus its row-block partition.
struct CsrView {
  const int *rowptr;
  const int *col;
  const double *val;
  const int *rowblk;  // nblk + 1 row boundaries
  int rows, cols, nblk;
  long nnz;
};

// Host-side: split rows into blocks of <= kNnzPerWg nonzeros (a longer row is alone in its block).
inline std::vector<int> build_rowblocks(const int *rowptr, int rows) {
  std::vector<int> rb;
  rb.push_back(0);
  int start = 0;
  while (start < rows) {
    int end = start;
    long base = rowptr[start];
    // also cap rows per block so every lane has at most a few rows to reduce
    while (end < rows && (rowptr[end + 1] - base) <= kNnzPerWg && (end - start) < 4 * kSpmvThreads) ++end;
    if (end == start) end = start + 1;  // single long row
    rb.push_back(end);
    start = end;
  }
  return rb;
}

// ---- epilogues -------------------------------------------------------------
// operator()(row, sum, acc) consumes one finished row; kPartial > 0 means the
// block reduces acc[] and stores kPartial partial results at partial[k*nblk + b].

struct EpiStore {  // y[r] = s   or  y[r] += s
  double *y;
  int accumulate;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { y[r] = accumulate ? y[r] + s : s; }
};

struct EpiDivR {  // z[r] = s / ry[r]            (CG step a: z = R_y^{-1} A p)
  double *z;
  const double *ry;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { z[r] = s / ry[r]; }
};

struct EpiGp {  // Gp[r] = (Pp)[r] + s + rx[r] p[r];  partial sum of p.Gp   (CG step b)
  double *Gp;
  const double *p, *rx;
  int has_P;
  double *partial;
  static constexpr int kSums = 1, kMaxs = 0;
  __device__ void operator()(int r, double s, double *sums, double *) const {
    const double pr = p[r];
    double g = s + rx[r] * pr;
    if (has_P) g += Gp[r];
    Gp[r] = g;
    sums[0] += pr * g;
  }
};

struct EpiRhs {  // b_x[r] = rx_part[r] + s          (rhs: r_x + A' R_y^{-1} r_y)
  double *out;
  const double *add;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { out[r] = add[r] + s; }
};

struct EpiY {  // y[r] = s / ry[r] + vy[r]        (y = R_y^{-1}(A x - r_y), r_y = -R_y v_y)
  double *y;
  const double *ry, *vy;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { y[r] = s / ry[r] + vy[r]; }
};

// primal residual pieces at a convergence check:
//  s = A x;  ax_s = s + sl;  ax_s_btau = ax_s - b tau.  Norms in normalised and original (/(D sigma)) space.
struct EpiResPri {
  const double *slack, *b, *Dinv;  // Dinv[r] = 1/(D[r] sigma) or nullptr
  const double *tau_ptr;           // |u[l-1]|
  const double *y;                 // dual iterate, for b'y
  double *partial;
  static constexpr int kSums = 1, kMaxs = 5;
  __device__ void operator()(int r, double ax, double *sums, double *maxs) const {
    const double tau = fabs(*tau_ptr);
    const double sl = slack[r];
    const double ax_s = ax + sl, ax_s_btau = ax_s - b[r] * tau;
    const double f = Dinv ? Dinv[r] : 1.0;
    maxs[0] = fmax(maxs[0], abs_nan_inf(ax_s_btau));      // normalised ||Ax+s-b tau||
    maxs[1] = fmax(maxs[1], abs_nan_inf(ax_s_btau * f));  // original
    maxs[2] = fmax(maxs[2], abs_nan_inf(ax_s * f));
    maxs[3] = fmax(maxs[3], abs_nan_inf(ax * f));
    maxs[4] = fmax(maxs[4], abs_nan_inf(sl * f));
    sums[0] += b[r] * y[r];
  }
};

// dual residual pieces: aty = A'y; px_aty_ctau = px + aty + c tau
struct EpiResDual {
  const double *px, *c, *Einv, *x;
  const double *tau_ptr;
  double *partial;
  static constexpr int kSums = 2, kMaxs = 4;
  __device__ void operator()(int r, double aty, double *sums, double *maxs) const {
    const double tau = fabs(*tau_ptr);
    const double pxr = px ? px[r] : 0.0;
    const double tot = pxr + aty + c[r] * tau;
    const double f = Einv ? Einv[r] : 1.0;
    maxs[0] = fmax(maxs[0], abs_nan_inf(tot));
    maxs[1] = fmax(maxs[1], abs_nan_inf(tot * f));
    maxs[2] = fmax(maxs[2], abs_nan_inf(pxr * f));
    maxs[3] = fmax(maxs[3], abs_nan_inf(aty * f));
    sums[0] += c[r] * x[r];
    sums[1] += pxr * x[r];
  }
};

template <class Epi>
__global__ __launch_bounds__(kSpmvThreads) void k_spmv_stream(CsrView A, const double *__restrict__ x, Epi epi,
                                                               const int *done_flag) {
  if (done_flag && *done_flag) return;
  __shared__ double prod[kNnzPerWg];
  __shared__ double red[kSpmvThreads / 64];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int r0 = A.rowblk[b], r1 = A.rowblk[b + 1];
  const int p0 = A.rowptr[r0], p1 = A.rowptr[r1];
  const int nnz = p1 - p0;
  constexpr int NS = Epi::kSums > 0 ? Epi::kSums : 1, NM = Epi::kMaxs > 0 ? Epi::kMaxs : 1;
  double sums[NS], maxs[NM];
#pragma unroll
  for (int i = 0; i < NS; ++i) sums[i] = 0.;
#pragma unroll
  for (int i = 0; i < NM; ++i) maxs[i] = 0.;

  if (nnz <= kNnzPerWg) {
    const double *__restrict__ v = A.val + p0;
    const int *__restrict__ c = A.col + p0;
#pragma unroll 8
    for (int k = tid; k < nnz; k += kSpmvThreads) prod[k] = v[k] * x[c[k]];
    __syncthreads();
    for (int r = r0 + tid; r < r1; r += kSpmvThreads) {
      const int a = A.rowptr[r] - p0, e = A.rowptr[r + 1] - p0;
      double s = 0.;
      for (int k = a; k < e; ++k) s += prod[k];
      epi(r, s, sums, maxs);
    }
  } else {  // one long row: whole workgroup reduces it (fixed order)
    double s = 0.;
    for (int k = p0 + tid; k < p1; k += kSpmvThreads) s += A.val[k] * x[A.col[k]];
    s = block_sum<kSpmvThreads>(s, red);
    if (tid == 0) epi(r0, s, sums, maxs);
  }
  if constexpr (Epi::kSums > 0 || Epi::kMaxs > 0) {
#pragma unroll
    for (int i = 0; i < Epi::kSums; ++i) {
      const double t = block_sum<kSpmvThreads>(sums[i], red);
      if (tid == 0) epi.partial[(size_t)i * gridDim.x + b] = t;
    }
#pragma unroll
    for (int i = 0; i < Epi::kMaxs; ++i) {
      const double t = block_max<kSpmvThreads>(maxs[i], red);
      if (tid == 0) epi.partial[(size_t)(Epi::kSums + i) * gridDim.x + b] = t;
    }
  }
}

template <class Epi>
inline void launch_spmv(const CsrView &A, const double *x, const Epi &epi, const int *done_flag, hipStream_t s) {
  if (A.nblk <= 0) return;
  hipLaunchKernelGGL(k_spmv_stream<Epi>, dim3(A.nblk), dim3(kSpmvThreads), 0, s, A, x, epi, done_flag);
}

}  // namespace scship
