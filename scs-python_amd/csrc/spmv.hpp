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
#include <algorithm>

#include "common.hpp"
#include "spmv_cs.hpp"

namespace scship {

constexpr int kSpmvThreads = 256;
constexpr int kNnzPerWg = 2048;  // 16 KiB of LDS products per workgroup
constexpr int kRowsPerLane = 4;  // a row block holds at most kRowsPerLane * kSpmvThreads rows

// Device view of a CSR matrix plus its row-block partition.
struct CsrView {
  const int *rowptr;
  const int *col;
  const double *val;
  const int4 *blk;    // per row block {first row, end row, first nonzero, end nonzero}: one load instead of a dependent pair
  int rows, cols, nblk;
  long nnz;
  int pstride = 0, pbase = 0;  // reduction partials: slot stride / first slot (0: this launch's own workgroup count / 0)
  int nlong = 0;               // k_spmv_peeled: the first nlong row blocks get a whole workgroup each
};

// Host-side: split rows into blocks of <= kNnzPerWg nonzeros (a longer row is alone in its block).
inline std::vector<int> build_rowblock_bounds(const int *rowptr, int rows) {
  std::vector<int> rb;
  rb.push_back(0);
  int start = 0;
  while (start < rows) {
    int end = start;
    long base = rowptr[start];
    // also cap rows per block so every lane has at most a few rows to reduce
    while (end < rows && (rowptr[end + 1] - base) <= kNnzPerWg && (end - start) < kRowsPerLane * kSpmvThreads) ++end;
    if (end == start) end = start + 1;  // single long row
    rb.push_back(end);
    start = end;
  }
  return rb;
}

inline std::vector<int4> build_rowblocks(const int *rowptr, int rows) {
  const std::vector<int> rb = build_rowblock_bounds(rowptr, rows);
  std::vector<int4> out(rb.size() - 1);
  for (size_t b = 0; b + 1 < rb.size(); ++b) out[b] = int4{rb[b], rb[b + 1], rowptr[rb[b]], rowptr[rb[b + 1]]};
  return out;
}

// A diagonal block of R = diag(diag_r) as the epilogues see it: the vector itself, or — inside the ADMM workspace, where
// R_x = rho_x I and R_y takes two values (zero-cone rows, the rest) — two scalars and the boundary: the same quotients
// and products bit for bit, without streaming 8 bytes per row for them.
struct RDiag {
  const double *vec = nullptr;
  double head = 0., rest = 0.;  // value of rows < nhead / of the others (vec == nullptr)
  int nhead = 0;
  RDiag() = default;
  __host__ __device__ RDiag(const double *v) : vec(v) {}
  __host__ __device__ RDiag(double head_, double rest_, int nhead_) : head(head_), rest(rest_), nhead(nhead_) {}
  __device__ __forceinline__ double operator[](int r) const { return vec ? vec[r] : (r < nhead ? head : rest); }
};

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
  RDiag ry;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { z[r] = s / ry[r]; }
};

struct EpiGp {  // Gp[r] = (Pp)[r] + s + rx[r] p[r];  partial sum of p.Gp   (CG step b)
  double *Gp;
  const double *p;
  RDiag rx;
  int has_P;
  double *partial;
  double *Gp2 = nullptr;  // split layouts (spmv_cs.hpp, two workgroups per row chunk): the second half's partial sums
  static constexpr int kSums = 1, kMaxs = 0;
  __device__ void operator()(int r, double s, double *sums, double *) const {
    const double pr = p[r];
    double g = s + rx[r] * pr;
    if (has_P) g += Gp[r];
    Gp[r] = g;
    sums[0] += pr * g;
  }
  // Round 4: the pass kernel (spmv_cs.hpp k_spmv_cs_il) fetches p[r] of a lane's rows at its START into registers and hands it
  // back here — the epilogue then has no load left in it (K2 ran 5 us behind K1, whose epilogue EpiDivR reads nothing: 8 dependent
  // HBM-latency loads per lane at the end of the launch, when nothing is left to overlap them with).  Same arithmetic, same bits.
  static constexpr bool kPrefetch = true;
  __device__ double prefetch(int r) const { return p[r]; }
  __device__ void with_prefetched(int r, double s, int split, int part, double pr, double *sums) const {
    if (split <= 1 || part == 0) {
      double g = s + rx[r] * pr;
      if (has_P) g += Gp[r];
      Gp[r] = g;
      sums[0] += pr * g;
    } else {
      Gp2[r] = s;
      sums[0] += pr * s;
    }
  }
  // Gp = Gp[] + Gp2[] is only ever read by the CG update (k_cg_update / k_cg_init), and p'Gp is linear in it: the
  // first half carries the R_x p (+ P p) terms, the second half its raw partial sum — no combine pass
  __device__ void split(int r, double s, int part, double *sums, double *) const {
    const double pr = p[r];
    if (part == 0) {
      double g = s + rx[r] * pr;
      if (has_P) g += Gp[r];
      Gp[r] = g;
      sums[0] += pr * g;
    } else {
      Gp2[r] = s;
      sums[0] += pr * s;
    }
  }
};

struct EpiPartial {  // split layouts: raw partial row sums of the two halves (finished by k_epi_finish)
  double *y0, *y1;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { y0[r] = s; }
  __device__ void split(int r, double s, int part, double *, double *) const { (part ? y1 : y0)[r] = s; }
};

// warm-started CG start, fused:  r0 = R_x (v_x - ws) - (P ws) - s  with s = A'(v_y + R_y^{-1} A ws);
// p0 = M r0; partials [sum r0 M r0 | max |r0|]
struct EpiR0 {
  double *r, *p;
  const double *M;
  RDiag rx;
  const double *vx, *ws, *Pws;  // Pws nullable
  double *partial;
  static constexpr int kSums = 1, kMaxs = 1;
  __device__ void operator()(int j, double s, double *sums, double *maxs) const {
    double r0 = rx[j] * (vx[j] - ws[j]) - s;
    if (Pws) r0 -= Pws[j];
    const double z = M[j] * r0;
    r[j] = r0;
    p[j] = z;
    sums[0] += z * r0;
    maxs[0] = fmax(maxs[0], abs_nan_inf(r0));
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
  RDiag ry;
  const double *vy;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { y[r] = s / ry[r] + vy[r]; }
};

// primal residual pieces at a convergence check:
//  ax = A x;  ax_s = ax + s;  ax_s_btau = ax_s - b tau.  Norms in normalised and original (/(D sigma)) space.
// sums: [b'y, ||ax_s_btau||_2^2 normalised, original]; maxs: see RES_P_* below
enum : int { RES_P_BTY = 0, RES_P_SQ_N, RES_P_SQ_O, RES_P_MAX_N, RES_P_MAX_O, RES_P_AXS_O, RES_P_AX_O, RES_P_S_O, RES_P_AXS_N,
             RES_P_COUNT };
struct EpiResPri {
  const double *slack, *b, *Dinv;  // Dinv[r] = 1/(D[r] sigma) or nullptr
  const double *tau_ptr;           // |u[l-1]|
  const double *y;                 // dual iterate, for b'y
  double *partial;
  static constexpr int kSums = 3, kMaxs = 6;
  __device__ void operator()(int r, double ax, double *sums, double *maxs) const {
    const double tau = fabs(*tau_ptr);
    const double sl = slack[r];
    const double ax_s = ax + sl, ax_s_btau = ax_s - b[r] * tau;
    const double f = Dinv ? Dinv[r] : 1.0;
    sums[0] += b[r] * y[r];
    sums[1] += ax_s_btau * ax_s_btau;
    sums[2] += (ax_s_btau * f) * (ax_s_btau * f);
    maxs[0] = fmax(maxs[0], abs_nan_inf(ax_s_btau));      // normalised ||Ax+s-b tau||
    maxs[1] = fmax(maxs[1], abs_nan_inf(ax_s_btau * f));  // original
    maxs[2] = fmax(maxs[2], abs_nan_inf(ax_s * f));
    maxs[3] = fmax(maxs[3], abs_nan_inf(ax * f));
    maxs[4] = fmax(maxs[4], abs_nan_inf(sl * f));
    maxs[5] = fmax(maxs[5], abs_nan_inf(ax_s));
  }
};

// dual residual pieces: aty = A'y; px_aty_ctau = px + aty + c tau
// sums: [c'x, x'Px, ||.||_2^2 normalised, original]; maxs: see RES_D_*
enum : int { RES_D_CTX = 0, RES_D_XPX, RES_D_SQ_N, RES_D_SQ_O, RES_D_MAX_N, RES_D_MAX_O, RES_D_PX_O, RES_D_ATY_O, RES_D_PX_N,
             RES_D_ATY_N, RES_D_COUNT };
struct EpiResDual {
  const double *px, *c, *Einv, *x;
  const double *tau_ptr;
  double *partial;
  static constexpr int kSums = 4, kMaxs = 6;
  __device__ void operator()(int r, double aty, double *sums, double *maxs) const {
    const double tau = fabs(*tau_ptr);
    const double pxr = px ? px[r] : 0.0;
    const double tot = pxr + aty + c[r] * tau;
    const double f = Einv ? Einv[r] : 1.0;
    sums[0] += c[r] * x[r];
    sums[1] += pxr * x[r];
    sums[2] += tot * tot;
    sums[3] += (tot * f) * (tot * f);
    maxs[0] = fmax(maxs[0], abs_nan_inf(tot));
    maxs[1] = fmax(maxs[1], abs_nan_inf(tot * f));
    maxs[2] = fmax(maxs[2], abs_nan_inf(pxr * f));
    maxs[3] = fmax(maxs[3], abs_nan_inf(aty * f));
    maxs[4] = fmax(maxs[4], abs_nan_inf(pxr));
    maxs[5] = fmax(maxs[5], abs_nan_inf(aty));
  }
};

// One row block of the CSR-stream mat-vec, executed by kSpmvThreads consecutive lanes (tid = 0..255 inside the
// group).  `nb` = number of row blocks of the whole product = stride of the epilogue's partial arrays.
// Shared by the one-launch-per-product kernel below and by the persistent CG kernel (cg_persist.hpp), so both
// paths produce the same bits.  sync() must synchronise the lanes that share `prod` / `red`.
// UNIFORM: several groups share one hardware barrier (persistent kernel), so every call — short-row block,
// long-row block or idle (`active` = false) — executes the same number of sync() calls.
template <class Epi, bool UNIFORM = false, class Sync>
__device__ __forceinline__ void spmv_stream_block(const CsrView &A, const double *__restrict__ x, const Epi &epi, int b, int nb,
                                                  double *prod, double *red, int tid, Sync sync, bool active = true) {
  if (UNIFORM && !active) {
#pragma unroll
    for (int i = 0; i < 3 + 2 * (Epi::kSums + Epi::kMaxs); ++i) sync();
    return;
  }
  const int4 bi = A.blk[b];
  const int r0 = bi.x, r1 = bi.y, p0 = bi.z, p1 = bi.w;
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
    // the row extents of this lane's (at most 4) rows do not depend on the products: fetch them together with
    // the nonzeros instead of after the barrier (one global round trip less on the critical path)
    int ra[kRowsPerLane], re[kRowsPerLane];
#pragma unroll
    for (int j = 0; j < kRowsPerLane; ++j) {
      const int r = r0 + tid + j * kSpmvThreads;
      ra[j] = r < r1 ? A.rowptr[r] - p0 : 0;
      re[j] = r < r1 ? A.rowptr[r + 1] - p0 : 0;
    }
#pragma unroll 8
    for (int k = tid; k < nnz; k += kSpmvThreads) prod[k] = v[k] * x[c[k]];
    sync();
#pragma unroll
    for (int j = 0; j < kRowsPerLane; ++j) {
      const int r = r0 + tid + j * kSpmvThreads;
      if (r < r1) {
        double s = 0.;
        for (int k = ra[j]; k < re[j]; ++k) s += prod[k];
        epi(r, s, sums, maxs);
      }
    }
    if (UNIFORM) { sync(); sync(); }
  } else {  // one long row: the whole group reduces it (fixed order)
    double s = 0.;
    for (int k = p0 + tid; k < p1; k += kSpmvThreads) s += A.val[k] * x[A.col[k]];
    if (UNIFORM) sync();
    s = group_sum<kSpmvThreads>(s, red, tid, sync);
    if (tid == 0) epi(r0, s, sums, maxs);
  }
  if constexpr (Epi::kSums > 0 || Epi::kMaxs > 0) {
#pragma unroll
    for (int i = 0; i < Epi::kSums; ++i) {
      const double t = group_sum<kSpmvThreads>(sums[i], red, tid, sync);
      if (tid == 0) epi.partial[(size_t)i * nb + b] = t;
    }
#pragma unroll
    for (int i = 0; i < Epi::kMaxs; ++i) {
      const double t = group_max<kSpmvThreads>(maxs[i], red, tid, sync);
      if (tid == 0) epi.partial[(size_t)(Epi::kSums + i) * nb + b] = t;
    }
  }
}

template <class Epi>
__device__ __forceinline__ void d_spmv_stream(CsrView A, const double *__restrict__ x, Epi epi, const int *done_flag,
                                              int *step_counter) {
  if (done_flag && *done_flag) return;
  if ((int)blockIdx.x >= A.nblk) return;  // (grouped launches, batch.hpp: the grid is sized for the problem with the most row blocks)
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;  // one CG step begins
  __shared__ double prod[kNnzPerWg];
  __shared__ double red[kSpmvThreads / 64];
  // (blocks are addressed by blockIdx; the partial slots may be offset: side launches of the column-sorted layouts)
  CsrView B = A;
  B.blk = A.blk - A.pbase;
  spmv_stream_block(B, x, epi, A.pbase + (int)blockIdx.x, A.pstride > 0 ? A.pstride : (int)gridDim.x, prod, red, (int)threadIdx.x, BlockSync{});
}
template <class Epi>
__global__ __launch_bounds__(kSpmvThreads) void k_spmv_stream(CsrView A, const double *__restrict__ x, Epi epi,
                                                               const int *done_flag, int *step_counter) {
  d_spmv_stream(A, x, epi, done_flag, step_counter);
}

// ---------------------------------------------------------------------------
// L2-blocked "slab" variant for matrices whose gather vector does not fit in one
// XCD's 4 MiB L2 (n ~ 1e6 => 8-16 MB of x).
//
// Measurements that shaped it (profiles/, devtools/gather_bench.hip): 2e7 random
// 8-byte gathers cost 72 us from a 128 KB table, 87 us from 1 MB (L2 hits),
// 177 us from 8 MB and 257 us from 16 MB (L2 misses served by the fabric, one
// 128-byte line per gather) — the texture-address path, not HBM, bounds this
// kernel: ~2 TA cycles per divergent lane, ~0.25 per coalesced lane.
//
// Design: column-block the matrix into S slabs of 2^kSlabShift columns (1 MiB of
// x) and make every workgroup own a fixed chunk of R = 256*RPT rows for the whole
// launch, walking the slabs in order s = 0..S-1.  The grid is sized to be fully
// resident (<= 4 workgroups per CU), every workgroup has the same amount of work
// per slab, so all of them advance through the slabs together and each XCD's L2
// only has to hold the current slab (+ a neighbour).  Row accumulators stay in
// REGISTERS across slabs — no partial-sum traffic.  Inside a (chunk, slab) segment:
// each lane streams 4 consecutive nonzeros with 16-byte loads (segments are padded
// to a multiple of 4), gathers x, stages the products in LDS, then sums the LDS
// runs of its RPT rows.  Slabs are ascending column ranges, so every row is still
// summed in ascending-column order: bit-identical to the CSR-stream kernel and to
// the oracle.  Extra bytes vs CSR: 2 B * rows * S of uint16 row offsets.
// ---------------------------------------------------------------------------
constexpr int kSlabShiftDefault = 17;        // 2^17 columns = 1 MiB of fp64 per slab
constexpr int kSlabStage = 3072;             // LDS products per pass (24 KiB)
constexpr int kSlabTargetWgs = 512;          // 2 resident workgroups per CU on 256 CUs
inline int slab_shift() {
  const int v = opts().slab_shift;  // (labs knob)
  return (v >= 10 && v <= 24) ? v : kSlabShiftDefault;
}

constexpr int kSlabRoffPad = 16;  // row offsets of a segment: R + 1 used, stride R + 16 ushorts (32-byte aligned rows of 16)
struct SlabView {
  const int *segptr;            // nchunks*S + 1 offsets into val/col (multiples of 4)
  const unsigned short *roff;   // (nchunks*S) x (R+16): R+1 row offsets inside a segment, padded
  const int *col;
  const double *val;
  int rows, cols, nchunks, S, R, max_seg;
};

struct HostSlab {
  std::vector<int> segptr, col;
  std::vector<unsigned short> roff;
  std::vector<double> val;
  int rows = 0, cols = 0, nchunks = 0, S = 0, R = 0, max_seg = 0;
};

inline int slab_pick_rows(int rows) {
  { const int v = opts().slab_rpt; if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) return 256 * v; }  // (labs knob)
  int rpt = 1;
  while (rpt < 16 && (long)kSlabTargetWgs * 256 * rpt < rows) rpt *= 2;
  return 256 * rpt;
}
inline bool slab_wanted(int rows, int cols) { return (long)cols * 8 > (2L << 20) && rows >= 65536; }
// The column-sorted pass layout (spmv_cs.hpp) pays as soon as there is a pass of nonzeros for most CUs — whether or
// not the gather vector fits L2 (what it saves is L2 -> L1 line traffic): measured crossover on LP+SOC problems with
// 10 nonzeros per row at nnz ~ 1e6 (0.240 vs 0.245 ms/iter), -10 % per iteration at nnz = 2e6, -12 % at 4e6.
inline bool cs_wanted(int rows, int cols, long nnz) {
  const long min_nnz = opts().cs_min_nnz;  // SCS_HIP_CS=N: tests on small matrices
  (void)cols;
  return rows >= 16384 && nnz >= min_nnz;
}

// CSR -> slab format; false when a (chunk, slab) segment would overflow the uint16 offsets
inline bool build_slab(const int *rowptr, const int *col, const double *val, int rows, int cols, HostSlab &out,
                       std::vector<int> *src = nullptr, int force_R = 0) {
  const int R = force_R > 0 ? force_R : slab_pick_rows(rows);
  const int shift = slab_shift();
  const int S = (int)(((long)cols + (1L << shift) - 1) >> shift);
  const int nchunks = (rows + R - 1) / R;
  const long nnz = rowptr[rows];
  out.rows = rows; out.cols = cols; out.R = R; out.S = S; out.nchunks = nchunks; out.max_seg = 0;
  out.segptr.assign((size_t)nchunks * S + 1, 0);
  out.roff.assign((size_t)nchunks * S * (R + kSlabRoffPad), 0);
  out.col.clear(); out.val.clear();
  if (src) { src->clear(); src->reserve(nnz + 4L * nchunks * S); }
  out.col.reserve(nnz + 4L * nchunks * S);
  out.val.reserve(nnz + 4L * nchunks * S);
  std::vector<int> cursor(R);
  long copied = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int r0 = c * R, r1 = std::min(rows, r0 + R);
    for (int r = r0; r < r1; ++r) cursor[r - r0] = rowptr[r];
    for (int s = 0; s < S; ++s) {
      const size_t seg = (size_t)c * S + s;
      const long seg0 = (long)out.col.size();
      if (seg0 > 2000000000L) return false;
      out.segptr[seg] = (int)seg0;
      unsigned short *ro = &out.roff[seg * (R + kSlabRoffPad)];
      const long chi = (long)(s + 1) << shift;  // first column beyond this slab
      for (int r = r0; r < r1; ++r) {
        const long off = (long)out.col.size() - seg0;
        if (off > 65535) return false;
        ro[r - r0] = (unsigned short)off;
        int p = cursor[r - r0];
        const int pe = rowptr[r + 1];
        while (p < pe && col[p] < chi) {
          out.col.push_back(col[p]);
          out.val.push_back(val[p]);
          if (src) src->push_back(p);
          ++p; ++copied;
        }
        cursor[r - r0] = p;
      }
      const long endoff = (long)out.col.size() - seg0;
      if (endoff > 65535) return false;
      for (int r = r1; r <= r0 + R; ++r) ro[r - r0] = (unsigned short)endoff;
      // pad to a multiple of 4 with zero-valued entries inside the slab (never summed: beyond every row's run)
      const int padcol = (int)std::min<long>((long)s << shift, (long)cols - 1);
      while (out.col.size() % 4 != 0) { out.col.push_back(padcol); out.val.push_back(0.0); if (src) src->push_back(-1); }
      out.max_seg = std::max(out.max_seg, (int)((long)out.col.size() - seg0));
    }
  }
  out.segptr[(size_t)nchunks * S] = (int)out.col.size();
  return copied == nnz;
}

// the RPT + 1 row offsets of lane tid (rows tid*RPT ...): 16-byte loads where the lane's run is 16-byte aligned
// (RPT = 8, 16: 1 or 2 uint4 + the closing offset) instead of RPT + 1 two-byte loads that each touch 16 lines
template <int RPT>
__device__ __forceinline__ void slab_load_offs(const unsigned short *__restrict__ seg_roff, int tid, int (&o)[RPT + 1]) {
  const unsigned short *ro = seg_roff + tid * RPT;
  if constexpr (RPT % 8 == 0) {
#pragma unroll
    for (int v = 0; v < RPT / 8; ++v) {
      const uint4 w = reinterpret_cast<const uint4 *>(ro)[v];
      o[8 * v + 0] = w.x & 0xffff; o[8 * v + 1] = w.x >> 16;
      o[8 * v + 2] = w.y & 0xffff; o[8 * v + 3] = w.y >> 16;
      o[8 * v + 4] = w.z & 0xffff; o[8 * v + 5] = w.z >> 16;
      o[8 * v + 6] = w.w & 0xffff; o[8 * v + 7] = w.w >> 16;
    }
    o[RPT] = ro[RPT];
  } else {
#pragma unroll
    for (int j = 0; j <= RPT; ++j) o[j] = ro[j];
  }
}

// register image of one pass: NQ quads (4 nonzeros each) per lane
template <int NQ>
struct SlabRegs {
  int4 cc[NQ];
  double2 va[NQ], vb[NQ];
};

template <int NQ>
__device__ __forceinline__ void slab_load(SlabRegs<NQ> &r, const int4 *__restrict__ c4, const double2 *__restrict__ v2,
                                          int cnt4, int tid) {
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = tid + i * kSpmvThreads;
    if (q < cnt4) {
      r.cc[i] = c4[q];
      r.va[i] = v2[2 * q];
      r.vb[i] = v2[2 * q + 1];
    }
  }
}
template <int NQ>
__device__ __forceinline__ void slab_gather(SlabRegs<NQ> &r, const double *__restrict__ x, int cnt4, int tid) {
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = tid + i * kSpmvThreads;
    if (q < cnt4) {
      r.va[i].x *= x[r.cc[i].x];
      r.va[i].y *= x[r.cc[i].y];
      r.vb[i].x *= x[r.cc[i].z];
      r.vb[i].y *= x[r.cc[i].w];
    }
  }
}
template <int NQ>
__device__ __forceinline__ void slab_stage(const SlabRegs<NQ> &r, double *prod, int cnt4, int tid) {
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = tid + i * kSpmvThreads;
    if (q < cnt4) {
      reinterpret_cast<double2 *>(prod)[2 * q] = r.va[i];
      reinterpret_cast<double2 *>(prod)[2 * q + 1] = r.vb[i];
    }
  }
}

template <class Epi, int RPT, int STAGE>
__global__ __launch_bounds__(kSpmvThreads) void k_spmv_slab(SlabView A, const double *__restrict__ x, Epi epi,
                                                             const int *done_flag, int *step_counter) {
  if (done_flag && *done_flag) return;
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;  // one CG step begins
  constexpr int R = kSpmvThreads * RPT;
  constexpr int NQ = STAGE / 4 / kSpmvThreads;
  __shared__ __attribute__((aligned(16))) double prod[STAGE];
  __shared__ double red[kSpmvThreads / 64];
  const int tid = threadIdx.x, c = blockIdx.x;
  constexpr int NS = Epi::kSums > 0 ? Epi::kSums : 1, NM = Epi::kMaxs > 0 ? Epi::kMaxs : 1;
  double sums[NS], maxs[NM], acc[RPT];
#pragma unroll
  for (int i = 0; i < NS; ++i) sums[i] = 0.;
#pragma unroll
  for (int i = 0; i < NM; ++i) maxs[i] = 0.;
#pragma unroll
  for (int j = 0; j < RPT; ++j) acc[j] = 0.;

  if (A.max_seg <= STAGE) {
    // ---- fast path: every segment fits one LDS pass; software-pipelined over the slabs:
    // the 16-byte val/col loads (HBM latency) and the row offsets of slab s+1 are issued right
    // after the gathers of slab s, so the texture-address path never waits on HBM.
    const size_t seg0 = (size_t)c * A.S;
    SlabRegs<NQ> cur, nxt;
    int o_cur[RPT + 1], o_nxt[RPT + 1];
    int p0 = A.segptr[seg0], p1 = A.segptr[seg0 + 1];
    int cnt4 = (p1 - p0) >> 2;
    slab_load<NQ>(cur, reinterpret_cast<const int4 *>(A.col + p0), reinterpret_cast<const double2 *>(A.val + p0), cnt4, tid);
    slab_load_offs<RPT>(A.roff + seg0 * (R + kSlabRoffPad), tid, o_cur);
    for (int s = 0; s < A.S; ++s) {
      slab_gather<NQ>(cur, x, cnt4, tid);
      int cnt4_n = 0;
      if (s + 1 < A.S) {
        const int q0 = p1, q1 = A.segptr[seg0 + s + 2];
        cnt4_n = (q1 - q0) >> 2;
        slab_load<NQ>(nxt, reinterpret_cast<const int4 *>(A.col + q0), reinterpret_cast<const double2 *>(A.val + q0), cnt4_n, tid);
        slab_load_offs<RPT>(A.roff + (seg0 + s + 1) * (R + kSlabRoffPad), tid, o_nxt);
        p1 = q1;
      }
      slab_stage<NQ>(cur, prod, cnt4, tid);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        double t = acc[j];
        for (int k = o_cur[j]; k < o_cur[j + 1]; ++k) t += prod[k];
        acc[j] = t;
      }
      __syncthreads();
      cur = nxt;
      cnt4 = cnt4_n;
#pragma unroll
      for (int j = 0; j <= RPT; ++j) o_cur[j] = o_nxt[j];
    }
  } else {
    // ---- general path: a segment may need several LDS passes
    for (int s = 0; s < A.S; ++s) {
      const size_t seg = (size_t)c * A.S + s;
      const int p0 = A.segptr[seg], n_seg = A.segptr[seg + 1] - p0;  // both multiples of 4
      int o[RPT + 1];
      slab_load_offs<RPT>(A.roff + seg * (R + kSlabRoffPad), tid, o);
      for (int base = 0; base < n_seg; base += STAGE) {
        const int cnt4 = min(STAGE, n_seg - base) >> 2;
        SlabRegs<NQ> cur;
        slab_load<NQ>(cur, reinterpret_cast<const int4 *>(A.col + p0 + base),
                      reinterpret_cast<const double2 *>(A.val + p0 + base), cnt4, tid);
        slab_gather<NQ>(cur, x, cnt4, tid);
        slab_stage<NQ>(cur, prod, cnt4, tid);
        __syncthreads();
        const int cnt = cnt4 << 2;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
          const int a = max(o[j], base) - base, e = min(o[j + 1], base + cnt) - base;
          double t = acc[j];
          for (int k = a; k < e; ++k) t += prod[k];
          acc[j] = t;
        }
        __syncthreads();
      }
    }
  }
  // Epilogue.  A lane owns RPT consecutive rows, so applying the epilogue functor directly would make every
  // access to its vectors (R_y, v, M, ... and the output) touch 64 different 128-byte lines per instruction —
  // measured +10 us (one vector) to +40 us (six vectors, EpiR0) per launch at m = 2e6.  The row sums go through
  // LDS (stride-17 padding: conflict-free both ways) so that lane t finishes rows t, t + 256, ...: coalesced.
  if constexpr (RPT > 1) {
    static_assert(R + R / 16 <= STAGE, "row sums are transposed through the product buffer");
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int rl = tid * RPT + j;
      prod[rl + (rl >> 4)] = acc[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int rl = j * kSpmvThreads + tid, r = c * R + rl;
      if (r < A.rows) epi(r, prod[rl + (rl >> 4)], sums, maxs);
    }
  } else {
    const int r = c * R + tid;
    if (r < A.rows) epi(r, acc[0], sums, maxs);
  }
  if constexpr (Epi::kSums > 0 || Epi::kMaxs > 0) {
#pragma unroll
    for (int i = 0; i < Epi::kSums; ++i) {
      const double t = block_sum<kSpmvThreads>(sums[i], red);
      if (tid == 0) epi.partial[(size_t)i * gridDim.x + c] = t;
    }
#pragma unroll
    for (int i = 0; i < Epi::kMaxs; ++i) {
      const double t = block_max<kSpmvThreads>(maxs[i], red);
      if (tid == 0) epi.partial[(size_t)(Epi::kSums + i) * gridDim.x + c] = t;
    }
  }
}

// Side launch of the column-sorted layouts: the rows peeled off the passes (longer than a count field holds).
// blk[i] = {row, row + 1, first nonzero, end}, the nlong longest rows (> kPeelLongRow nonzeros) first: those get a whole
// workgroup each, the others one WAVEFRONT each, four to a workgroup.  Lanes stride over the row with eight loads in
// flight, then a fixed shuffle tree.  (A row this long is no longer summed in the oracle's sequential order: 1e-13 rel.)
constexpr int kPeelRowsPerWg = kSpmvThreads / 64, kPeelLongRow = 2048;
inline int peel_wgs_for(int npeel, int nlong) { return nlong + (npeel - nlong + kPeelRowsPerWg - 1) / kPeelRowsPerWg; }
// PIECES (virtual rows of the column-sorted layout): blk[i] = {row, row + 1, first piece, end} and A.val = the piece sums the
// pass kernel left behind — the same tree over those instead of over products.
template <class Epi, bool PIECES = false>
__global__ __launch_bounds__(kSpmvThreads) void k_spmv_peeled(CsrView A, const double *__restrict__ x, Epi epi, const int *done_flag) {
  if (done_flag && *done_flag) return;
  __shared__ double red[kSpmvThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int NS = Epi::kSums > 0 ? Epi::kSums : 1, NM = Epi::kMaxs > 0 ? Epi::kMaxs : 1;
  double sums[NS], maxs[NM];
#pragma unroll
  for (int k = 0; k < NS; ++k) sums[k] = 0.;
#pragma unroll
  for (int k = 0; k < NM; ++k) maxs[k] = 0.;
  const bool whole_wg = (int)blockIdx.x < A.nlong;
  const int i = whole_wg ? (int)blockIdx.x : A.nlong + ((int)blockIdx.x - A.nlong) * kPeelRowsPerWg + (tid >> 6);
  const int step = whole_wg ? kSpmvThreads : 64, first = whole_wg ? tid : lane;
  double acc = 0.;
  int row = -1;
  if (i < A.nblk) {
    const int4 bi = A.blk[i];
    row = bi.x;
#pragma unroll 8
    for (int k = bi.z + first; k < bi.w; k += step) acc += PIECES ? A.val[k] : A.val[k] * x[A.col[k]];
  }
  if (whole_wg) {
    acc = block_sum<kSpmvThreads>(acc, red);
    __syncthreads();
    if (tid == 0) epi(row, acc, sums, maxs);
  } else {
    acc = wave_sum(acc);
    if (lane == 0 && row >= 0) epi(row, acc, sums, maxs);
  }
  if constexpr (Epi::kSums > 0 || Epi::kMaxs > 0) {
    const int nslot = A.pstride > 0 ? A.pstride : (int)gridDim.x, slot = A.pbase + (int)blockIdx.x;
#pragma unroll
    for (int k = 0; k < Epi::kSums; ++k) {
      const double t = block_sum<kSpmvThreads>(sums[k], red);
      if (tid == 0) epi.partial[(size_t)k * nslot + slot] = t;
    }
#pragma unroll
    for (int k = 0; k < Epi::kMaxs; ++k) {
      const double t = block_max<kSpmvThreads>(maxs[k], red);
      if (tid == 0) epi.partial[(size_t)(Epi::kSums + k) * nslot + slot] = t;
    }
  }
}

// One matrix, whichever layout init chose for it.  nblk = number of workgroups = number of
// reduction partials an epilogue writes.
struct SpmvMat {
  CsrView csr{};
  SlabView slab{};
  CsView cs{};
  bool use_slab = false, use_cs = false;  // use_cs wins (spmv_cs.hpp: column-sorted passes)
  double *part0 = nullptr, *part1 = nullptr;  // cs.split == 2: scratch for the partial row sums of epilogues without split()
  // rows peeled off the column-sorted layout (too long for its count fields): one row block each over the plain CSR,
  // done by a CSR-stream launch right behind the main one (bit r of cs.peel marks them)
  const int4 *peel_blk = nullptr;
  int npeel = 0, nlong = 0;
  bool cs_combine() const { return use_cs && cs.split > 1 && cs.ticket != nullptr; }
  int cs_main_wgs() const { return cs_combine() ? cs.nchunks : cs.nchunks * cs.split; }
  int peel_wgs() const { return peel_wgs_for(npeel, nlong); }
  int nblk() const { return use_cs ? cs_main_wgs() + peel_wgs() : use_slab ? slab.nchunks : csr.nblk; }
};

// split layouts: the epilogue of a product whose functor is not linear in the row sum — rows finished from the two
// partial vectors, gridDim.x = the number of workgroups of the product (= the stride of the reduction partials)
constexpr int kEpiFinishThreads = 1024;  // few workgroups (their count is the stride of the partials): many lanes each
template <class Epi>
__global__ __launch_bounds__(kEpiFinishThreads) void k_epi_finish(const double *__restrict__ y0, const double *__restrict__ y1, int rows, Epi epi,
                                                              const int *done_flag, const unsigned *__restrict__ peel, int pstride) {
  if (done_flag && *done_flag) return;
  __shared__ double red[kEpiFinishThreads / 64];
  constexpr int NS = Epi::kSums > 0 ? Epi::kSums : 1, NM = Epi::kMaxs > 0 ? Epi::kMaxs : 1;
  double sums[NS], maxs[NM];
#pragma unroll
  for (int i = 0; i < NS; ++i) sums[i] = 0.;
#pragma unroll
  for (int i = 0; i < NM; ++i) maxs[i] = 0.;
  for (long r = (long)blockIdx.x * kEpiFinishThreads + threadIdx.x; r < rows; r += (long)gridDim.x * kEpiFinishThreads)
    if (!cs_is_peeled(peel, (int)r)) epi((int)r, y0[r] + y1[r], sums, maxs);  // (peeled rows: the side launch runs their epilogue)
  if constexpr (Epi::kSums > 0 || Epi::kMaxs > 0) {
    const int nslot = pstride > 0 ? pstride : (int)gridDim.x;
#pragma unroll
    for (int i = 0; i < Epi::kSums; ++i) {
      const double t = block_sum<kEpiFinishThreads>(sums[i], red);
      if (threadIdx.x == 0) epi.partial[(size_t)i * nslot + blockIdx.x] = t;
    }
#pragma unroll
    for (int i = 0; i < Epi::kMaxs; ++i) {
      const double t = block_max<kEpiFinishThreads>(maxs[i], red);
      if (threadIdx.x == 0) epi.partial[(size_t)(Epi::kSums + i) * nslot + blockIdx.x] = t;
    }
  }
}

// reduction partials an epilogue leaves per workgroup (kSums + kMaxs): the callers' partial buffers hold this many per workgroup
constexpr int kMaxEpiReductions = 10;  // EpiResDual: 4 sums + 6 maxima
template <class Epi>
inline void launch_spmv(const SpmvMat &M, const double *x, const Epi &epi, const int *done_flag, hipStream_t s,
                        int *step_counter = nullptr) {
  static_assert(Epi::kSums + Epi::kMaxs <= kMaxEpiReductions, "partial buffers are sized for kMaxEpiReductions values per workgroup");
  if (M.use_cs) {
    if (M.cs.nchunks <= 0) return;
    CsView V = M.cs;
    const int nmain = M.cs_main_wgs();
    if (M.npeel > 0) V.pstride = nmain + M.peel_wgs();  // the side launch's partials follow the main launch's
    bool finished = false;
    if (!M.cs_combine()) {  // (combine mode: partial sums are added inside the kernel, finished rows for any epilogue)
      if constexpr (!epi_has_split<Epi>::value) {
        if (M.cs.split > 1) {
          launch_spmv_cs(V, x, EpiPartial{M.part0, M.part1}, done_flag, s, step_counter);
          hipLaunchKernelGGL(k_epi_finish<Epi>, dim3(nmain), dim3(kEpiFinishThreads), 0, s, M.part0, M.part1, M.cs.rows, epi, done_flag,
                             M.cs.peel, V.pstride);
          finished = true;
        }
      }
    }
    if (!finished) launch_spmv_cs(V, x, epi, done_flag, s, step_counter);
    if (M.npeel > 0) {  // the peeled rows, whole, through the plain epilogue (EpiGp: Gp2 stays 0 for them)
      CsrView S = M.csr;
      S.blk = M.peel_blk;
      S.nblk = M.npeel;
      S.nlong = M.nlong;
      S.pstride = nmain + M.peel_wgs();
      S.pbase = nmain;
      if (M.cs.npieces > 0) {  // virtual rows: the long rows' nonzeros rode in the passes; add up their pieces
        S.val = M.cs.tpart;
        hipLaunchKernelGGL((k_spmv_peeled<Epi, true>), dim3(M.peel_wgs()), dim3(kSpmvThreads), 0, s, S, x, epi, done_flag);
      } else {
        hipLaunchKernelGGL((k_spmv_peeled<Epi, false>), dim3(M.peel_wgs()), dim3(kSpmvThreads), 0, s, S, x, epi, done_flag);
      }
    }
    return;
  }
  if (M.use_slab) {
    if (M.slab.nchunks <= 0) return;
    const dim3 g(M.slab.nchunks), b(kSpmvThreads);
    if (M.slab.R == 256) hipLaunchKernelGGL((k_spmv_slab<Epi, 1, kSlabStage>), g, b, 0, s, M.slab, x, epi, done_flag, step_counter);
    else if (M.slab.R == 512) hipLaunchKernelGGL((k_spmv_slab<Epi, 2, kSlabStage>), g, b, 0, s, M.slab, x, epi, done_flag, step_counter);
    else if (M.slab.R == 1024) hipLaunchKernelGGL((k_spmv_slab<Epi, 4, kSlabStage>), g, b, 0, s, M.slab, x, epi, done_flag, step_counter);
    else if (M.slab.R == 2048) hipLaunchKernelGGL((k_spmv_slab<Epi, 8, kSlabStage>), g, b, 0, s, M.slab, x, epi, done_flag, step_counter);
    else hipLaunchKernelGGL((k_spmv_slab<Epi, 16, 2 * kSlabStage>), g, b, 0, s, M.slab, x, epi, done_flag, step_counter);
    return;
  }
  if (M.csr.nblk <= 0) return;
  hipLaunchKernelGGL(k_spmv_stream<Epi>, dim3(M.csr.nblk), dim3(kSpmvThreads), 0, s, M.csr, x, epi, done_flag, step_counter);
}

}  // namespace scship
