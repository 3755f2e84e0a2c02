// spmv_cs.hpp — K1/K2 for large matrices: column-sorted passes with LDS-staged row partials.
//
// Same role as the slab kernel of spmv.hpp (scs_source/linsys/scs_matrix.c accum_by_a / accum_by_atrans, named at
// R:meson.build:199-202), different data layout.  What bounds a random-pattern SpMV on MI355X is the L2 -> L1 fill
// port of a CU: every 8-byte gather moves a 128-byte line (~2 clk); a wave instruction whose 64 lanes touch 64
// different lines costs 85 us per 2e7 gathers, one touching 24 lines 40 us (tools/gather_flavours.hip).  Lanes that
// own ROWS always gather 64 unrelated columns.  Here the lanes of an instruction take 64 CONSECUTIVE nonzeros of a
// column-sorted stream, so they share lines: the expected number of distinct lines is 64 * (1 - exp(-d)) / d with
// d = (nonzeros of the workgroup's rows per 128-byte line of x) = R * nnz_per_row * 16 / cols — it only depends on
// the number of rows R a workgroup owns, hence ONE workgroup of 1024 lanes per CU owning R = 1024 * RPT rows.
//
// Layout: rows are cut into chunks of R; the nonzeros of a chunk, sorted by (column, row), are cut into passes of
// kCsPass = 8192.  A pass stores  val[8192] (fp64),  idx[8192] (uint32 = column relative to the pass's first column
// << 13 | slot)  and  meta[1024] (uint64 per lane: first slot of the lane's run | RPT counts).  `slot` is the rank of
// the nonzero in (owner lane, row-of-lane j, column) order (cs_row_key): products are scattered to LDS by slot,
// then every lane sums the LDS runs of its RPT rows (row = chunk * R + j * 1024 + lane: epilogue accesses are
// coalesced) into REGISTER accumulators that live across the passes.  Passes are ascending column ranges and a run is in ascending column
// order, so every row is summed in ascending-column order from 0.0 — bit-identical to the CSR-stream kernel, the
// slab kernel and the oracle's loops.  The LDS product buffer is double-buffered: one barrier per pass.
// Inside a block of 256 consecutive sorted nonzeros the storage order is lane-major (lane l holds sorted positions
// l, 64 + l, 128 + l, 192 + l as one 16-byte quad), so a lane reads 16-byte pieces while each of its four gather
// instructions still covers 64 consecutive sorted nonzeros.
// Bytes streamed per nonzero: 8 + 4 + 1 (meta) = 13 (slab kernel: 12 + ~1.6 of row offsets).
#pragma once
#include <algorithm>
#include <cstdint>
#include <type_traits>

#include "common.hpp"

namespace scship {

constexpr int kCsThreads = 1024;
constexpr int kCsPass = 8192;                 // nonzeros per pass = LDS products per buffer (64 KiB)
constexpr int kCsQuads = kCsPass / 4 / kCsThreads;  // 16-byte quads per lane and pass
constexpr int kCsSlotBits = 13;
constexpr int kCsColBits = 32 - kCsSlotBits;  // a pass may span 2^19 columns
constexpr int kCsTargetWgs = 256;             // one resident workgroup per CU

struct CsView {
  const int *passptr;          // nchunks * split + 1: first pass of every workgroup (chunk c, part j: entry c * split + j)
  const int2 *pinfo;           // per pass {first column, nonzeros (the rest of the kCsPass entries is padding)}
  const unsigned *idx;         // npass * kCsPass
  const double *val;           // npass * kCsPass
  const unsigned long long *meta;  // npass * kCsThreads * cs_meta_words(rpt)
  int rows, cols, nchunks, R, npass, rpt;  // R rows per chunk (<= 1024 * rpt), rpt = accumulators per lane (1, 2, 4, 8, 16)
  int split;                   // workgroups per chunk (1, 2, 4: the chunk's column-sorted stream cut into equal parts; partial row sums)
  // In-kernel combine (k_spmv_cs_il only; nullptr = the workgroups hand their partial row sums to the epilogue's split()):
  // every workgroup of a chunk publishes its partial sums in `scratch` and takes a ticket; the LAST arriver adds the
  // parts in fixed order and runs the epilogue on the finished rows.  No spinning: nothing ever waits for a workgroup.
  double *scratch = nullptr;   // nchunks * split * 1024 * rpt doubles
  unsigned *ticket = nullptr;  // nchunks counters, monotone: split arrivals per launch
  // Peeled rows: rows with more nonzeros than a run-descriptor count holds are left out of the passes (bit r of `peel`)
  // and done by a CSR-stream side launch on the plain CSR (spmv.hpp launch_spmv); their epilogue is skipped here.
  const unsigned *peel = nullptr;
  int pstride = 0, pbase = 0;  // reduction partials: slot stride / first slot (0: this launch's own workgroup count / 0)
  // Virtual rows (round 3): the R row slots of a chunk are Rr REAL rows (chunk c: rows c * Rr ...) followed by Rp PIECE slots
  // (pieces c * Rp ...).  A row too long for the count fields has no nonzeros in its own slot (its `peel` bit is set: no epilogue
  // here); its nonzeros are dealt round-robin — the k-th of the row to piece k mod np — to np piece slots, which spreads every
  // piece over all passes of its chunk (a piece holds 1/np of what the row has in any column range) and the pieces over all
  // chunks.  A piece slot's sum goes to tpart[piece]; spmv.hpp launch_spmv adds a row's pieces (fixed wave tree) and runs the
  // epilogue on the finished row.  Plain layouts: Rr = R, Rp = 0.
  int Rr = 0, Rp = 0, npieces = 0;
  double *tpart = nullptr;
};
__host__ __device__ inline int cs_rr(const CsView &A) { return A.Rr > 0 ? A.Rr : A.R; }
// slot (row-local index rl of chunk c) -> what the finished sum is: a real row's (epilogue) or a piece's (stored)
template <class Epi>
__device__ __forceinline__ void cs_slot_done(const CsView &A, const Epi &epi, int part, int c, int rl, double s, double *sums, double *maxs,
                                             bool finished);
__host__ __device__ inline bool cs_is_peeled(const unsigned *peel, int r) { return peel && ((peel[r >> 5] >> (r & 31)) & 1u); }

// bits per row count in a run descriptor; 16 rows per lane use two words (8 counts each)
__host__ __device__ inline int cs_count_bits(int rpt) { return rpt == 16 ? 6 : (48 / rpt < 13 ? 48 / rpt : 13); }
__host__ __device__ inline int cs_meta_words(int rpt) { return rpt == 16 ? 2 : 1; }
// longest row the passes take: what a count field holds, and not more than one CSR-stream LDS stage
__host__ __device__ inline int cs_peel_threshold(int rpt) {
  const int mx = (1 << cs_count_bits(rpt)) - 1;
  return mx < 2048 ? mx : 2048;
}

// Chunk geometry: R rows per workgroup so that the launch is ONE wave of workgroups on the 256 CUs (every CU busy,
// as many rows per CU as possible: the distinct lines per gather instruction fall with R), R a multiple of 64;
// rpt = the power of two >= R / 1024.  More than 256 * 16384 rows: R = 16384 and several rounds of workgroups.
inline void cs_pick_geometry(int rows, int &R, int &rpt, int split = 1) {
  { const int v = opts().cs_rpt; if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) { rpt = v; R = kCsThreads * v; return; } }  // (labs knob: full chunks of 1024 * rpt rows)
  const int chunks = kCsTargetWgs / split;
  long r = ((long)rows + chunks - 1) / chunks;
  r = std::max(64L, (r + 63) / 64 * 64);
  r = std::min<long>(r, 16L * kCsThreads);
  rpt = 1;
  while ((long)kCsThreads * rpt < r) rpt *= 2;
  R = (int)r;
}

struct HostCs {
  std::vector<int> passptr;
  std::vector<int2> pinfo;
  std::vector<unsigned> idx;
  std::vector<double> val;
  std::vector<unsigned long long> meta;
  int rows = 0, cols = 0, nchunks = 0, R = 0, npass = 0, rpt = 0, split = 1;
};

// Passes of one workgroup are balanced: its n nonzeros go into ceil(n / kCsPass) passes of (nearly) equal length, a
// multiple of 256, instead of full passes and one short one — the kernel processes whole passes (padding included),
// so the padding is what this bounds: < 256 slots per pass.
__host__ __device__ inline int cs_pass_len(long n) {
  const long np = (n + kCsPass - 1) / kCsPass;
  if (np <= 0) return kCsPass;
  const long len = (((n + np - 1) / np) + 255) & ~255L;
  return (int)(len < kCsPass ? len : kCsPass);
}

// first stream position of part k of a chunk with n nonzeros cut into `split` parts (k = split: n)
__host__ __device__ inline long cs_part_cut(long n, int k, int split) {
  if (k <= 0) return 0;
  if (k >= split) return n;
  const long c = ((n * k) / split + 255) & ~255L;
  return c < n ? c : n;
}

// Slot order inside a pass: (owner lane, the lane's j-th row, column); row-local index rl = j * 1024 + lane.
// (Measured alternative: rows of a wave interleaved — (wave, j, lane) — with a DPP prefix sum per row so that the
// row-sum reads of neighbouring lanes are bank-conflict free: not faster, 104 vs 99 us on the K1 shape.)
__host__ __device__ inline int cs_row_key(int rl, int rpt) { return (rl & (kCsThreads - 1)) * rpt + rl / kCsThreads; }

// storage position of sorted position q inside a pass (lane-major quads inside blocks of 256)
__host__ __device__ inline int cs_store_pos(int q) {
  const int blk = q >> 8, k = (q >> 6) & 3, lane = q & 63;
  return (blk << 8) | (lane << 2) | k;
}

// Host builder (tests, lab, fallback): CSR -> column-sorted passes.  false when the pattern does not fit the
// format's bit fields (a pass wider than 2^19 columns, or more nonzeros of one row in one pass than the count
// field holds): the caller keeps the slab / CSR-stream layout.
inline bool build_cs(const int *rowptr, const int *col, const double *val, int rows, int cols, HostCs &out, int force_rpt = 0,
                     int split = 1, const unsigned *peel = nullptr, int force_R = 0) {
  int R, rpt;
  cs_pick_geometry(rows, R, rpt, split);
  if (force_rpt > 0) { rpt = force_rpt; R = force_R > 0 ? force_R : kCsThreads * rpt; }
  const int nchunks = (rows + R - 1) / R;
  const int cb = cs_count_bits(rpt);
  // longest run of one row inside one pass: what the count field holds, and not more than 2048 — the lane that owns the row adds its
  // run sequentially (a 70 000-nonzero budget row left whole made every pass of its chunk wait 8191 dependent adds: a solve of 6 s
  // took 87 s, tools/dbg/dense_rows_solve.py); longer rows are cut into pieces (CsView::Rr) or peeled by the caller
  const unsigned maxcnt = (unsigned)cs_peel_threshold(rpt);
  out.rows = rows; out.cols = cols; out.R = R; out.rpt = rpt; out.nchunks = nchunks; out.split = split;
  out.passptr.assign((size_t)nchunks * split + 1, 0);
  out.pinfo.clear(); out.idx.clear(); out.val.clear(); out.meta.clear();
  const int RK = kCsThreads * rpt;  // row keys: lane * rpt + j
  std::vector<int> cnt(RK), start(RK), run(RK);
  out.passptr[0] = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int r0 = c * R, r1 = std::min(rows, r0 + R);
    struct Ent { int col, rl, p; };
    std::vector<Ent> ents;
    ents.reserve((size_t)(rowptr[r1] - rowptr[r0]));
    for (int r = r0; r < r1; ++r) {
      if (cs_is_peeled(peel, r)) continue;  // done by the side launch
      for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) ents.push_back(Ent{col[p], r - r0, p});
    }
    std::sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.col != b.col ? a.col < b.col : a.rl < b.rl; });
    const long n_all = (long)ents.size();
    // split > 1: the chunk's stream is cut into `split` parts at (multiples of 256 near) k * n / split, one workgroup each
    for (int part = 0; part < split; ++part) {
    const long e_lo = cs_part_cut(n_all, part, split), n = cs_part_cut(n_all, part + 1, split);
    int np = 0;
    const int plen = cs_pass_len(n - e_lo);
    for (long e0 = e_lo, e1; e0 < n; e0 = e1, ++np) {
      // a pass = up to plen (<= kCsPass) consecutive sorted nonzeros spanning fewer than 2^kCsColBits columns
      e1 = std::min(n, e0 + plen);
      const int base = ents[e0].col;
      while ((long)ents[e1 - 1].col - base >= (1L << kCsColBits)) --e1;
      std::fill(cnt.begin(), cnt.end(), 0);
      auto keyof = [&](int rl) { return cs_row_key(rl, rpt); };
      for (long e = e0; e < e1; ++e) cnt[keyof(ents[e].rl)]++;
      int acc = 0;
      for (int k = 0; k < RK; ++k) { start[k] = acc; acc += cnt[k]; run[k] = 0; if ((unsigned)cnt[k] > maxcnt) return false; }
      const size_t o = out.val.size();
      out.val.resize(o + kCsPass, 0.0);
      out.idx.resize(o + kCsPass, 0u);
      out.pinfo.push_back(int2{base, (int)(e1 - e0)});
      for (long e = e0; e < e0 + kCsPass; ++e) {
        const int q = (int)(e - e0), sp = cs_store_pos(q);
        if (e < e1) {
          const int k = keyof(ents[e].rl);
          const unsigned slot = (unsigned)(start[k] + run[k]++);
          out.val[o + sp] = val[ents[e].p];
          out.idx[o + sp] = ((unsigned)(ents[e].col - base) << kCsSlotBits) | slot;
        } else {
          out.idx[o + sp] = (unsigned)q;  // padding: zero value, its own slot beyond every run
        }
      }
      for (int t = 0; t < kCsThreads; ++t) {
        unsigned long long w = (unsigned long long)start[cs_row_key(t, rpt)], w1 = 0;  // first slot of the lane's run
        for (int j = 0; j < rpt; ++j) {
          const unsigned long long n_j = (unsigned long long)cnt[cs_row_key(j * kCsThreads + t, rpt)];
          if (j < 8 || rpt < 16) w |= n_j << (16 + cb * j);
          else w1 |= n_j << (cb * (j - 8));
        }
        out.meta.push_back(w);
        if (rpt == 16) out.meta.push_back(w1);
      }
    }
    out.passptr[(size_t)c * split + part + 1] = out.passptr[(size_t)c * split + part] + np;
    }
  }
  out.npass = out.passptr[(size_t)nchunks * split];
  // too many short passes (very wide, very sparse chunks): the padding would be streamed on every product
  if ((long)out.npass * kCsPass > rowptr[rows] + rowptr[rows] / 4 + (long)nchunks * split * kCsPass) return false;
  return true;
}


// ---- virtual rows (CsView::Rr): planning and the host builder ----
__host__ __device__ inline int cs_slot_of_row(int r, int Rr, int R) { return (r / Rr) * R + r % Rr; }
__host__ __device__ inline int cs_slot_of_piece(int p, int Rr, int Rp, int R) { return (p / Rp) * R + Rr + p % Rp; }
struct CsVirtPlan {
  int R = 0, rpt = 0, nchunks = 0, Rr = 0, Rp = 0, V = 0;
  long long_nnz = 0;
  std::vector<int2> rowinfo;   // per row {first piece, pieces} or {-1, 0}
  std::vector<int4> blk;       // per long row {row, row + 1, first piece, end}
  std::vector<unsigned> mask;  // bit r: long row (no epilogue in the pass kernel)
};
// rows longer than long_thresh nonzeros -> ceil(len / lp) pieces; the geometry of the slot space.  false: nothing to cut / too big
inline bool cs_plan_virtual(const int *rp, int rows, int lp, int long_thresh, CsVirtPlan &P) {
  P = CsVirtPlan{};
  P.rowinfo.assign((size_t)rows, int2{-1, 0});
  P.mask.assign(((size_t)rows + 31) / 32, 0u);
  long V = 0;
  for (int r = 0; r < rows; ++r) {
    const int len = rp[r + 1] - rp[r];
    if (len <= long_thresh) continue;
    const int np = (len + lp - 1) / lp;
    P.rowinfo[(size_t)r] = int2{(int)V, np};
    P.blk.push_back(int4{r, r + 1, (int)V, (int)V + np});
    P.mask[(size_t)r >> 5] |= 1u << (r & 31);
    V += np;
    P.long_nnz += len;
    if (V > 500000000L) return false;
  }
  if (P.blk.empty()) return false;
  P.V = (int)V;
  const long total = (long)rows + V;
  if (total > 2000000000L) return false;
  cs_pick_geometry((int)total, P.R, P.rpt, 1);
  for (;;) {  // every chunk: Rr real-row slots + Rp piece slots
    P.nchunks = (int)((total + P.R - 1) / P.R);
    P.Rp = (P.V + P.nchunks - 1) / P.nchunks;
    P.Rr = P.R - P.Rp;
    if (P.Rr >= 1 && (long)P.nchunks * P.Rr >= rows) break;
    P.R += 64;
    while (P.R > kCsThreads * P.rpt) P.rpt *= 2;
    if (P.rpt > 16) return false;
  }
  return true;
}
// the layout of the slot space: a CSR over nchunks * R slots (a row's nonzeros in ascending column order keep that order inside
// every piece), then build_cs with the plan's geometry.  out.rows = the REAL row count.
inline bool build_cs_virtual(const int *rp, const int *ci, const double *v, int rows, int cols, const CsVirtPlan &P, HostCs &out) {
  const long slots = (long)P.nchunks * P.R, nnz = rp[rows];
  std::vector<int> vrp((size_t)slots + 1, 0), vci((size_t)nnz);
  std::vector<double> vv((size_t)nnz);
  auto slot_of = [&](int r, int k) {
    const int2 info = P.rowinfo[(size_t)r];
    return info.x < 0 ? cs_slot_of_row(r, P.Rr, P.R) : cs_slot_of_piece(info.x + k % info.y, P.Rr, P.Rp, P.R);
  };
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < rp[r + 1] - rp[r]; ++k) vrp[(size_t)slot_of(r, k) + 1]++;
  for (long i = 0; i < slots; ++i) vrp[(size_t)i + 1] += vrp[(size_t)i];
  std::vector<int> fill(vrp.begin(), vrp.end() - 1);
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < rp[r + 1] - rp[r]; ++k) {
      const int d = fill[(size_t)slot_of(r, k)]++;
      vci[(size_t)d] = ci[rp[r] + k];
      vv[(size_t)d] = v[rp[r] + k];
    }
  if (!build_cs(vrp.data(), vci.data(), vv.data(), (int)slots, cols, out, P.rpt, 1, nullptr, P.R)) return false;
  out.rows = rows;
  return true;
}

// split == 1: the finished row goes through the epilogue functor.  split == 2: `s` is the partial sum of the
// workgroup's half of the row — only epilogues that are linear in it (a `split` member: EpiGp, EpiPartial) may be
// launched on such a layout (launch_spmv routes the others through EpiPartial + k_epi_finish).
template <class Epi, class = void>
struct epi_has_split : std::false_type {};
template <class Epi>
struct epi_has_split<Epi, std::void_t<decltype(&Epi::split)>> : std::true_type {};
template <class Epi>
__device__ __forceinline__ void cs_epilogue(const Epi &epi, int split, int part, int r, double s, double *sums, double *maxs) {
  if constexpr (epi_has_split<Epi>::value) {
    if (split > 1) { epi.split(r, s, part, sums, maxs); return; }
  }
  epi(r, s, sums, maxs);
}

template <class Epi>
__device__ __forceinline__ void cs_slot_done(const CsView &A, const Epi &epi, int part, int c, int rl, double s, double *sums, double *maxs,
                                             bool finished) {
  const int Rr = cs_rr(A);
  if (rl < Rr) {
    const int r = c * Rr + rl;
    if (r < A.rows && !cs_is_peeled(A.peel, r)) {
      if (finished) epi(r, s, sums, maxs);  // (in-kernel combine: the plain epilogue on finished rows)
      else cs_epilogue(epi, A.split, part, r, s, sums, maxs);
    }
  } else if (rl < A.R) {
    const int p = c * A.Rp + (rl - Rr);
    if (p < A.npieces) A.tpart[p] = s;
  }
}

// epilogues that want a per-row value fetched ahead of time (EpiGp: p[r]; spmv.hpp)
template <class Epi, class = void>
struct epi_has_prefetch : std::false_type {};
template <class Epi>
struct epi_has_prefetch<Epi, std::void_t<decltype(&Epi::prefetch)>> : std::true_type {};
// the row behind slot rl of chunk c if its finished sum goes through the epilogue HERE (not a piece slot, not peeled), else -1
__device__ __forceinline__ int cs_epilogue_row(const CsView &A, int c, int rl) {
  const int Rr = cs_rr(A);
  if (rl >= Rr) return -1;
  const int r = c * Rr + rl;
  return (r < A.rows && !cs_is_peeled(A.peel, r)) ? r : -1;
}

// Row sums of one pass from the LDS product buffer: m = {first slot of the lane's run | the lane's RPT counts}.
// (Measured alternative: level by level — the L-th product of several rows as one batch of independent LDS reads,
// +0.0 for rows without one — is not faster than the plain per-row loops: 100-102 vs 99 us on the K1 shape.)
template <int RPT>
__device__ __forceinline__ void cs_row_sums(const double *__restrict__ pb, unsigned long long m, unsigned long long m1, double (&acc)[RPT]) {
  constexpr int CB = RPT == 16 ? 6 : (48 / RPT < 13 ? 48 / RPT : 13);
  constexpr unsigned CM = (1u << CB) - 1;
  int o = (int)(m & 0xffff);
  unsigned long long w = m >> 16;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    if (RPT == 16 && j == 8) w = m1;
    const int n = (int)((unsigned)w & CM);
    w >>= CB;
    double t = acc[j];
    for (int k = 0; k < n; ++k) t += pb[o + k];
    acc[j] = t;
    o += n;
  }
}

template <class Epi, int RPT>
__global__ __launch_bounds__(kCsThreads) void k_spmv_cs(CsView A, const double *__restrict__ x, Epi epi, const int *done_flag,
                                                         int *step_counter) {
  if (done_flag && *done_flag) return;
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;  // one CG step begins
  constexpr int NQ = kCsQuads;
  __shared__ __attribute__((aligned(16))) double prod[2][kCsPass];
  __shared__ double red[kCsThreads / 64];
  const int tid = threadIdx.x, wg = blockIdx.x, c = wg / A.split, part = wg - c * A.split;
  constexpr int NS = Epi::kSums > 0 ? Epi::kSums : 1, NM = Epi::kMaxs > 0 ? Epi::kMaxs : 1;
  double sums[NS], maxs[NM], acc[RPT];
#pragma unroll
  for (int i = 0; i < NS; ++i) sums[i] = 0.;
#pragma unroll
  for (int i = 0; i < NM; ++i) maxs[i] = 0.;
#pragma unroll
  for (int j = 0; j < RPT; ++j) acc[j] = 0.;

  const int g0 = A.passptr[wg], g1 = A.passptr[wg + 1];
  uint4 ic[NQ], in[NQ];
  double2 va[NQ], vb[NQ], na[NQ], nb[NQ];
  unsigned long long mc = 0, mn = 0, mc1 = 0, mn1 = 0;
  int2 pc{0, 0}, pn{0, 0};
  // a lane's i-th quad belongs to the block of 256 sorted nonzeros (tid >> 6) + 16 i: blocks beyond the pass's
  // nonzero count hold only padding and are skipped (wave-uniform)
  auto load = [&](int g, uint4(&ii)[NQ], double2(&a)[NQ], double2(&b)[NQ], unsigned long long &m, unsigned long long &m1, int2 &pi) {
    const uint4 *i4 = reinterpret_cast<const uint4 *>(A.idx + (size_t)g * kCsPass);
    const double2 *v2 = reinterpret_cast<const double2 *>(A.val + (size_t)g * kCsPass);
    pi = A.pinfo[g];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * kCsThreads;
      if (((q >> 6) << 8) < pi.y) {
        ii[i] = i4[q];
        a[i] = v2[2 * q];
        b[i] = v2[2 * q + 1];
      }
    }
    m = A.meta[((size_t)g * kCsThreads + tid) * (RPT == 16 ? 2 : 1)];
    if (RPT == 16) m1 = A.meta[((size_t)g * kCsThreads + tid) * 2 + 1];
  };
  if (g0 < g1) load(g0, ic, va, vb, mc, mc1, pc);
  int buf = 0;
  for (int g = g0; g < g1; ++g) {
    // gathers of this pass first, then the streaming loads of the next one: the in-order return queue hands the
    // gathered x back without waiting for the HBM latency of the stream
    const double *xb = x + pc.x;
    double xg[NQ][4];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if ((((tid + i * kCsThreads) >> 6) << 8) < pc.y) {
        xg[i][0] = xb[ic[i].x >> kCsSlotBits];
        xg[i][1] = xb[ic[i].y >> kCsSlotBits];
        xg[i][2] = xb[ic[i].z >> kCsSlotBits];
        xg[i][3] = xb[ic[i].w >> kCsSlotBits];
      }
    }
    if (g + 1 < g1) load(g + 1, in, na, nb, mn, mn1, pn);
    double *pb = prod[buf];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if ((((tid + i * kCsThreads) >> 6) << 8) < pc.y) {
        pb[ic[i].x & (kCsPass - 1)] = va[i].x * xg[i][0];
        pb[ic[i].y & (kCsPass - 1)] = va[i].y * xg[i][1];
        pb[ic[i].z & (kCsPass - 1)] = vb[i].x * xg[i][2];
        pb[ic[i].w & (kCsPass - 1)] = vb[i].y * xg[i][3];
      }
    }
    __syncthreads();
    cs_row_sums<RPT>(pb, mc, mc1, acc);
    buf ^= 1;
#pragma unroll
    for (int i = 0; i < NQ; ++i) { ic[i] = in[i]; va[i] = na[i]; vb[i] = nb[i]; }
    mc = mn;
    mc1 = mn1;
    pc = pn;
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    cs_slot_done(A, epi, part, c, j * kCsThreads + tid, acc[j], sums, maxs, false);
  }
  if constexpr (Epi::kSums > 0 || Epi::kMaxs > 0) {
    const int nslot = A.pstride > 0 ? A.pstride : (int)gridDim.x, slot = A.pbase + wg;
#pragma unroll
    for (int i = 0; i < Epi::kSums; ++i) {
      const double t = block_sum<kCsThreads>(sums[i], red);
      if (tid == 0) epi.partial[(size_t)i * nslot + slot] = t;
    }
#pragma unroll
    for (int i = 0; i < Epi::kMaxs; ++i) {
      const double t = block_max<kCsThreads>(maxs[i], red);
      if (tid == 0) epi.partial[(size_t)(Epi::kSums + i) * nslot + slot] = t;
    }
  }
}


// Gather-ahead schedule of the same pass loop (same data, same per-row order => same bits): the gathers of pass
// g + 1 are issued right after the barrier of pass g, BEFORE its LDS row sums, and the streaming loads of pass g + 2
// behind them, so the vector-memory pipe works through the LDS / barrier phase and every streamed pass has a whole
// pass of time to arrive.  Two register sets alternate (the loop is unrolled by two: no copies of in-flight loads).
template <int NQ>
struct CsSet {
  uint4 ic[NQ];
  double2 va[NQ], vb[NQ];
  unsigned long long meta, meta1;
  int2 pi;
};

// ABL (tools/slab_lab.hip ablations, results intentionally wrong): 1 = gathers folded into a 2 KB table,
// 2 = no LDS row sums, 3 = no LDS traffic at all, 4 = no streamed values (idx only)
template <class Epi, int RPT, int ABL = 0>
__global__ __launch_bounds__(kCsThreads) void k_spmv_cs_ga(CsView A, const double *__restrict__ x, Epi epi, const int *done_flag,
                                                            int *step_counter) {
  if (done_flag && *done_flag) return;
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;  // one CG step begins
  constexpr int NQ = kCsQuads;
  __shared__ __attribute__((aligned(16))) double prod[2][kCsPass];
  __shared__ double red[kCsThreads / 64];
  const int tid = threadIdx.x, wg = blockIdx.x, c = wg / A.split, part = wg - c * A.split;
  constexpr int NS = Epi::kSums > 0 ? Epi::kSums : 1, NM = Epi::kMaxs > 0 ? Epi::kMaxs : 1;
  double sums[NS], maxs[NM], acc[RPT];
#pragma unroll
  for (int i = 0; i < NS; ++i) sums[i] = 0.;
#pragma unroll
  for (int i = 0; i < NM; ++i) maxs[i] = 0.;
#pragma unroll
  for (int j = 0; j < RPT; ++j) acc[j] = 0.;
  const int g0 = A.passptr[wg], g1 = A.passptr[wg + 1];
  CsSet<NQ> S0, S1;
  double xg[NQ][4];
  auto load = [&](int g, CsSet<NQ> &S) {
    const uint4 *i4 = reinterpret_cast<const uint4 *>(A.idx + (size_t)g * kCsPass);
    const double2 *v2 = reinterpret_cast<const double2 *>(A.val + (size_t)g * kCsPass);
    S.pi = A.pinfo[g];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * kCsThreads;
      if (((q >> 6) << 8) < S.pi.y) {
        S.ic[i] = i4[q];
        if (ABL == 4) { S.va[i] = double2{1., 1.}; S.vb[i] = double2{1., 1.}; }
        else { S.va[i] = v2[2 * q]; S.vb[i] = v2[2 * q + 1]; }
      }
    }
    S.meta = A.meta[((size_t)g * kCsThreads + tid) * (RPT == 16 ? 2 : 1)];
    if (RPT == 16) S.meta1 = A.meta[((size_t)g * kCsThreads + tid) * 2 + 1];
  };
  auto gather = [&](const CsSet<NQ> &S) {
    const double *xb = x + S.pi.x;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if ((((tid + i * kCsThreads) >> 6) << 8) < S.pi.y) {
        constexpr unsigned GM = ABL == 1 ? 255u : 0xffffffffu;
        xg[i][0] = xb[(S.ic[i].x >> kCsSlotBits) & GM];
        xg[i][1] = xb[(S.ic[i].y >> kCsSlotBits) & GM];
        xg[i][2] = xb[(S.ic[i].z >> kCsSlotBits) & GM];
        xg[i][3] = xb[(S.ic[i].w >> kCsSlotBits) & GM];
      }
    }
  };
  // pass g lives in X (its gathers are in flight), pass g + 1 streams into Y
  auto step = [&](int g, CsSet<NQ> &X, CsSet<NQ> &Y, int buf) {
    double *pb = prod[buf];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if ((((tid + i * kCsThreads) >> 6) << 8) < X.pi.y) {
        if (ABL == 3) {
          acc[0] += X.va[i].x * xg[i][0] + X.va[i].y * xg[i][1] + X.vb[i].x * xg[i][2] + X.vb[i].y * xg[i][3];
        } else {
          pb[X.ic[i].x & (kCsPass - 1)] = X.va[i].x * xg[i][0];
          pb[X.ic[i].y & (kCsPass - 1)] = X.va[i].y * xg[i][1];
          pb[X.ic[i].z & (kCsPass - 1)] = X.vb[i].x * xg[i][2];
          pb[X.ic[i].w & (kCsPass - 1)] = X.vb[i].y * xg[i][3];
        }
      }
    }
    const unsigned long long mc = X.meta, mc1 = RPT == 16 ? X.meta1 : 0;
    __syncthreads();
    if (g + 1 < g1) gather(Y);
    if (g + 2 < g1) load(g + 2, X);
    if (ABL == 2 || ABL == 3) { acc[0] += (double)(mc & 0xffffff); return; }
    cs_row_sums<RPT>(pb, mc, mc1, acc);
  };
  if (g0 < g1) {
    load(g0, S0);
    gather(S0);
    if (g0 + 1 < g1) load(g0 + 1, S1);
  }
  for (int g = g0; g < g1; g += 2) {
    step(g, S0, S1, 0);
    if (g + 1 < g1) step(g + 1, S1, S0, 1);
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    cs_slot_done(A, epi, part, c, j * kCsThreads + tid, acc[j], sums, maxs, false);
  }
  if constexpr (Epi::kSums > 0 || Epi::kMaxs > 0) {
    const int nslot = A.pstride > 0 ? A.pstride : (int)gridDim.x, slot = A.pbase + wg;
#pragma unroll
    for (int i = 0; i < Epi::kSums; ++i) {
      const double t = block_sum<kCsThreads>(sums[i], red);
      if (tid == 0) epi.partial[(size_t)i * nslot + slot] = t;
    }
#pragma unroll
    for (int i = 0; i < Epi::kMaxs; ++i) {
      const double t = block_max<kCsThreads>(maxs[i], red);
      if (tid == 0) epi.partial[(size_t)(Epi::kSums + i) * nslot + slot] = t;
    }
  }
}

// Braided schedule (default): same passes, same per-row order => same bits as k_spmv_cs_ga.  What the timeline of
// the gather-ahead kernel shows (tools/cs_lab.hip, s_memtime stamps): a wave spends 43 % of the launch ISSUING the
// gathers of the next pass — the texture addresser takes ~2.3 clk per distinct 128-byte line and back-pressures the
// issue — then 20 % in the LDS row sums of the current pass while the addresser runs dry, then waits at the barrier for
// the slowest issuer.  Here the two are braided: after the barrier every lane alternates "issue a slice of the memory
// instructions of the coming passes" / "sum one of its rows from LDS", so the LDS latency hides behind the addresser
// instead of following it.  Two details keep hipcc's wait-count insertion exact (a conditional load in one arm of a
// branch makes it fall back to vmcnt(0), i.e. to draining the stream loads it just issued): full passes — all but the
// last of a workgroup — run a branch-free body with unconditional loads (MASKED = false), and the per-pass header
// {first column, count} comes through the scalar cache (constant address space) two passes ahead.
typedef const int __attribute__((address_space(4))) *CsPinfoScalarPtr;

// tools/cs_lab.hip defines CS_LAB_TIMELINE (and the buffer) to get per-wave phase sums out of the kernel; no-ops otherwise
#ifdef CS_LAB_TIMELINE
#define CS_TL_DECL unsigned long long tl_[6] = {0, 0, 0, 0, 0, 0}, tl_t0 = __builtin_amdgcn_s_memtime(), tl_t1; const unsigned long long tl_begin = tl_t0
#define CS_TL_STAMP(slot) do { tl_t1 = __builtin_amdgcn_s_memtime(); tl_[slot] += tl_t1 - tl_t0; tl_t0 = tl_t1; } while (0)
#define CS_TL_FLUSH() do { if ((threadIdx.x & 63) == 0) { unsigned long long *o_ = cs_lab_tl + ((size_t)blockIdx.x * (kCsThreads / 64) + (threadIdx.x >> 6)) * 8; \
    for (int i_ = 0; i_ < 6; ++i_) o_[i_] = tl_[i_]; o_[6] = tl_t1 - tl_begin; } } while (0)
#else
#define CS_TL_DECL
#define CS_TL_STAMP(slot)
#define CS_TL_FLUSH()
#endif

template <class Epi, int RPT, int ABL = 0>
__global__ __launch_bounds__(kCsThreads) void k_spmv_cs_il(CsView A, const double *__restrict__ x, Epi epi, const int *done_flag,
                                                            int *step_counter) {
  if (done_flag && *done_flag) return;
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;  // one CG step begins
  constexpr int NQ = kCsQuads;
  constexpr int CB = RPT == 16 ? 6 : (48 / RPT < 13 ? 48 / RPT : 13);
  constexpr unsigned CM = (1u << CB) - 1;
  __shared__ __attribute__((aligned(16))) double prod[2][kCsPass];
  __shared__ double red[kCsThreads / 64];
  __shared__ unsigned ticket_sm;
  const int tid = threadIdx.x, wg = blockIdx.x, c = wg / A.split, part = wg - c * A.split;
  constexpr int NS = Epi::kSums > 0 ? Epi::kSums : 1, NM = Epi::kMaxs > 0 ? Epi::kMaxs : 1;
  double sums[NS], maxs[NM], acc[RPT];
#pragma unroll
  for (int i = 0; i < NS; ++i) sums[i] = 0.;
#pragma unroll
  for (int i = 0; i < NM; ++i) maxs[i] = 0.;
#pragma unroll
  for (int j = 0; j < RPT; ++j) acc[j] = 0.;
  const int g0 = A.passptr[wg], g1 = A.passptr[wg + 1];
  CS_TL_DECL;
  // (the default schedule only: the register budget of the others is not worth touching)
  constexpr bool kPre = epi_has_prefetch<Epi>::value && ABL == 6;
  double pre[kPre ? RPT : 1];
  bool pre_ok = false;
  if (g0 < g1) {
    pre_ok = kPre;
    CsPinfoScalarPtr pinf = (CsPinfoScalarPtr)A.pinfo;
    auto get_pi = [&](int g) {  // uniform index: s_load_dwordx2
      const int gg = g < g1 ? g : g1 - 1;
      return int2{pinf[2 * gg], pinf[2 * gg + 1]};
    };
    // Every load is unconditional: the padding of a pass holds zero values with their own slots beyond every run and
    // column offset 0, so it is harmless to process (build_cs balances the passes of a workgroup: < 2 % padding).
    // Issue order inside a braid: the gathers of pass g + 1 FIRST, the stream loads of pass g + 2 behind them — a
    // wave's loads return in order, and gathered lines that had to wait in the 32 KB L1 behind an HBM-latency load
    // of the same wave would be evicted before they are consumed.
    struct Set { uint4 ic[NQ]; double2 va[NQ], vb[NQ]; unsigned long long meta, meta1; };
    Set S0, S1;
    double xg[NQ][4];
    // (lab ablation 5, tools/cs_lab.hip: the pass stream with the non-temporal cache policy — does keeping the HBM-latency stream lines
    //  out of the L1 / L2 allocation help the gathers?  Round 4: no, profiles/r04_cs_stream_lab.txt)
    typedef unsigned cs_u4 __attribute__((ext_vector_type(4)));
    typedef double cs_f2 __attribute__((ext_vector_type(2)));
    auto ld_idx = [&](int s, Set &S, int gl) {
      if constexpr (ABL == 5) {
        const cs_u4 t = __builtin_nontemporal_load(reinterpret_cast<const cs_u4 *>(A.idx + (size_t)gl * kCsPass) + (tid + s * kCsThreads));
        S.ic[s] = uint4{t.x, t.y, t.z, t.w};
      } else {
        S.ic[s] = reinterpret_cast<const uint4 *>(A.idx + (size_t)gl * kCsPass)[tid + s * kCsThreads];
      }
    };
    auto ld_val = [&](int s, Set &S, int gl) {  // s in [0, 2 NQ)
      const int i = s >> 1, h = s & 1;
      double2 v;
      if constexpr (ABL == 5) {
        const cs_f2 t = __builtin_nontemporal_load(reinterpret_cast<const cs_f2 *>(A.val + (size_t)gl * kCsPass) + (2 * (tid + i * kCsThreads) + h));
        v = double2{t.x, t.y};
      } else {
        v = reinterpret_cast<const double2 *>(A.val + (size_t)gl * kCsPass)[2 * (tid + i * kCsThreads) + h];
      }
      if (h == 0) S.va[i] = v; else S.vb[i] = v;
    };
    auto ld_meta = [&](Set &S, int gl) {
      if constexpr (RPT == 16) {  // two words: one 16-byte load
        const ulonglong2 m2 = reinterpret_cast<const ulonglong2 *>(A.meta)[(size_t)gl * kCsThreads + tid];
        S.meta = m2.x;
        S.meta1 = m2.y;
      } else {
        S.meta = A.meta[(size_t)gl * kCsThreads + tid];
      }
    };
    auto gat = [&](int k, const Set &S, int col0) {  // k in [0, 4 NQ)
      const int i = k >> 2, e = k & 3;
      const unsigned id = e == 0 ? S.ic[i].x : e == 1 ? S.ic[i].y : e == 2 ? S.ic[i].z : S.ic[i].w;
      constexpr unsigned GM = ABL == 1 ? 255u : 0xffffffffu;  // (lab ablation 1: gathers folded into a 2 KB table)
      // (uniform base + unsigned 32-bit byte offset: the saddr form of global_load, no 64-bit address arithmetic)
      const unsigned off = ((id >> kCsSlotBits) & GM) << 3;
      xg[i][e] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x + col0) + off);
    };
    // memory instruction k of the braid of step g, in issue order: the gathers of pass g + 1 (in Y, first column c1),
    // then index quads, values and run descriptor of pass g + 2 into X.  TAIL: the last two steps of the workgroup,
    // where some of these passes do not exist (uniform branches).
    // PRE (lab ablation 6, round 4): the stream loads of pass g + 2 are issued BEFORE the barrier of step g — into the registers the
    // product scatter has just freed — instead of behind the gathers in the braid: while the workgroup waits at the barrier for its
    // slowest wavefront nobody issues memory instructions and the vector-memory pipe drains (the barrier is 26 % of a wavefront's
    // time, tools/cs_lab.hip timeline); loads that depend on nothing can fill that hole.
    // PREG quads of the gathers of pass g + 1 go before the barrier too, each right behind the products that free its registers
    // (7: one of the two quads, 9: both — then the braid issues nothing); 8: stream before the barrier, ALL gathers in the first row slot.
    constexpr bool PRE = (ABL >= 6 && ABL <= 10) || ABL == 12 || ABL == 14;   // (10: as 6 with ONE gather per row slot; 12: as 6 with straight-line tail steps; 13: plain braid with them)
    constexpr int PREG = ABL == 7 ? 1 : ABL == 9 ? NQ : 0;
    constexpr int NGAT = 4 * NQ, NVAL = 2 * NQ, NMEM = PRE ? NGAT : NGAT + NQ + NVAL + 1;
    // `tail` is a compile-time mode: 0 = passes g + 1 and g + 2 exist (steady state), 1 = test at run time (uniform branches — every
    // load inside one costs the exact wait counts), 2 = pass g + 1 exists, g + 2 does not, 3 = neither (the workgroup's last pass)
#define CS_HAS1 (TM == 0 || TM == 2 || (TM == 1 && g + 1 < g1))
#define CS_HAS2 (TM == 0 || (TM == 1 && g + 2 < g1))
    auto mem_op = [&](auto tail, int k, int g, const Set &Y, int c1, Set &X) {
      constexpr int TM = (int)decltype(tail)::value;
      if (k < 4 * PREG) return;
      if (k < NGAT) { if (CS_HAS1) gat(k, Y, c1); }
      else if (k < NGAT + NQ) { if (CS_HAS2) ld_idx(k - NGAT, X, g + 2); }
      else if (k < NGAT + NQ + NVAL) { if (CS_HAS2) ld_val(k - NGAT - NQ, X, g + 2); }
      else { if (CS_HAS2) ld_meta(X, g + 2); }
    };
    // products of pass g (in X) -> LDS, barrier, then the braid with the row sums of pass g
    auto step = [&](auto tail, int g, Set &X, const Set &Y, int c1, int buf) {
      constexpr int TM = (int)decltype(tail)::value;
      double *pb = prod[buf];
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        pb[X.ic[i].x & (kCsPass - 1)] = X.va[i].x * xg[i][0];
        pb[X.ic[i].y & (kCsPass - 1)] = X.va[i].y * xg[i][1];
        pb[X.ic[i].z & (kCsPass - 1)] = X.vb[i].x * xg[i][2];
        pb[X.ic[i].w & (kCsPass - 1)] = X.vb[i].y * xg[i][3];
        if constexpr (PREG > 0) {
          if (i < PREG && CS_HAS1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gat(4 * i + e, Y, c1);
          }
        }
      }
      const unsigned long long mc = X.meta, mc1 = RPT == 16 ? X.meta1 : 0;
      if constexpr (PRE) {
        if (CS_HAS2) {
#pragma unroll
          for (int s = 0; s < NQ; ++s) ld_idx(s, X, g + 2);
#pragma unroll
          for (int s = 0; s < NVAL; ++s) ld_val(s, X, g + 2);
          ld_meta(X, g + 2);
        }
      }
      CS_TL_STAMP(1);
      __syncthreads();
      CS_TL_STAMP(2);
      int o = (int)(mc & 0xffff);
      unsigned long long w = mc >> 16;
      // memory instructions per row: at least two, so that the gathers are all with the addresser after a few rows
      constexpr int PER = ABL == 8 ? NMEM : ABL == 10 ? (NMEM + RPT - 1) / RPT : (NMEM + RPT - 1) / RPT > 2 ? (NMEM + RPT - 1) / RPT : 2;
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
#pragma unroll
        for (int k = j * PER; k < (j + 1) * PER && k < NMEM; ++k) mem_op(tail, k, g, Y, c1, X);
        __builtin_amdgcn_sched_barrier(0);
        if (RPT == 16 && j == 8) w = mc1;
        const int n = (int)((unsigned)w & CM);
        w >>= CB;
        double t = acc[j];
        if (ABL == 2) t += (double)n;  // (lab ablation 2: no LDS row sums)
        else {
#pragma nounroll
          for (int k = 0; k < n; ++k) t += pb[o + k];  // (n is ~1: unrolling only costs registers)
        }
        acc[j] = t;
        o += n;
        __builtin_amdgcn_sched_barrier(0);
      }
      CS_TL_STAMP(3);
    };
    // first column of pass g + 1: scalar loads ahead of use
    int cn = get_pi(g0 + 1).x;
    if constexpr (ABL == 14) {  // (lab, round 4: the first gathers as early as possible — index quads, gathers, THEN the values of pass g0)
#pragma unroll
      for (int s = 0; s < NQ; ++s) ld_idx(s, S0, g0);
      const int c0 = get_pi(g0).x;
#pragma unroll
      for (int k = 0; k < NGAT; ++k) gat(k, S0, c0);
#pragma unroll
      for (int s = 0; s < NVAL; ++s) ld_val(s, S0, g0);
      ld_meta(S0, g0);
      if (g0 + 1 < g1) {
#pragma unroll
        for (int s = 0; s < NQ; ++s) ld_idx(s, S1, g0 + 1);
#pragma unroll
        for (int s = 0; s < NVAL; ++s) ld_val(s, S1, g0 + 1);
        ld_meta(S1, g0 + 1);
      }
    } else {  // prologue: stream passes g0 and g0 + 1, gather g0
#pragma unroll
      for (int s = 0; s < NQ; ++s) ld_idx(s, S0, g0);
#pragma unroll
      for (int s = 0; s < NVAL; ++s) ld_val(s, S0, g0);
      ld_meta(S0, g0);
      const int c0 = get_pi(g0).x;
#pragma unroll
      for (int k = 0; k < NGAT; ++k) gat(k, S0, c0);
      if (g0 + 1 < g1) {
#pragma unroll
        for (int s = 0; s < NQ; ++s) ld_idx(s, S1, g0 + 1);
#pragma unroll
        for (int s = 0; s < NVAL; ++s) ld_val(s, S1, g0 + 1);
        ld_meta(S1, g0 + 1);
      }
    }
    if constexpr (kPre) {
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const int r = cs_epilogue_row(A, c, j * kCsThreads + tid);
        pre[j] = epi.prefetch(r >= 0 ? r : 0);
      }
    }
    CS_TL_STAMP(0);
    // The steady-state loop holds ONLY the branch-free body (a variant with conditional loads inside the same loop
    // would merge its wait-count state into it); a second loop drains the last two or three passes.
    int g = g0;
    for (; g + 3 < g1; g += 2) {
      step(std::false_type{}, g, S0, S1, cn, 0);      // pass g in S0, g + 1 in S1, g + 2 streams into S0
      cn = get_pi(g + 2).x;
      step(std::false_type{}, g + 1, S1, S0, cn, 1);  // pass g + 1 in S1, g + 2 in S0, g + 3 streams into S1
      cn = get_pi(g + 3).x;
    }
    if constexpr (ABL == 12 || ABL == 13) {
      // the last two or three passes: one straight-line step body per case instead of run-time tests around every load
      using M0 = std::integral_constant<int, 0>; using M2 = std::integral_constant<int, 2>; using M3 = std::integral_constant<int, 3>;
      for (; g < g1; g += 2) {
        const int rem = g1 - g;
        if (rem >= 3) step(M0{}, g, S0, S1, cn, 0);
        else if (rem == 2) step(M2{}, g, S0, S1, cn, 0);
        else step(M3{}, g, S0, S1, cn, 0);
        cn = get_pi(g + 2).x;
        if (rem >= 4) step(M0{}, g + 1, S1, S0, cn, 1);
        else if (rem == 3) step(M2{}, g + 1, S1, S0, cn, 1);
        else if (rem == 2) step(M3{}, g + 1, S1, S0, cn, 1);
        cn = get_pi(g + 3).x;
      }
    } else {
    for (; g < g1; g += 2) {
      step(std::true_type{}, g, S0, S1, cn, 0);
      cn = get_pi(g + 2).x;
      if (g + 1 < g1) step(std::true_type{}, g + 1, S1, S0, cn, 1);
      cn = get_pi(g + 3).x;
    }
    }
  }
#undef CS_HAS1
#undef CS_HAS2
  const bool combine = A.split > 1 && A.ticket != nullptr;
  if (combine) {
    // Publish the partial row sums with write-through (sc1) 16-byte stores — acknowledged stores are visible to every
    // XCD, no L2 write-back fence needed, and 8-byte sc1 stores would cost one fabric write each — then one ticket per
    // workgroup; the last of the chunk's `split` arrivals finishes the rows.  Scratch order: pairs of a lane's rows.
    double *mine = A.scratch + (size_t)wg * (kCsThreads * RPT);
    if constexpr (RPT >= 2) {
      typedef double cs_d2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int j = 0; j < RPT; j += 2) {
        cs_d2 v;
        v.x = acc[j];
        v.y = acc[j + 1];
        cs_d2 *dst = reinterpret_cast<cs_d2 *>(mine) + (size_t)(j >> 1) * kCsThreads + tid;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
      }
    } else {
      __hip_atomic_store(mine + tid, acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ticket_sm = __hip_atomic_fetch_add(A.ticket + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket_sm % (unsigned)A.split != (unsigned)A.split - 1) return;
    // parts in fixed order 0 .. split - 1 (whoever combines): ((p0 + p1) + p2) + p3; the others' parts through
    // agent-scope (sc1) loads, which bypass this CU's L1
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      double t = 0.;
      for (int p = 0; p < A.split; ++p) {
        const size_t o = RPT >= 2 ? ((size_t)(j >> 1) * kCsThreads + tid) * 2 + (j & 1) : (size_t)tid;
        const double v = p == part ? acc[j]
                                   : __hip_atomic_load(A.scratch + (size_t)(c * A.split + p) * (kCsThreads * RPT) + o, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        t = p == 0 ? v : t + v;
      }
      acc[j] = t;
    }
  }
  if constexpr (kPre) {
    if (pre_ok && !combine) {
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const int rl = j * kCsThreads + tid, r = cs_epilogue_row(A, c, rl);
        if (r >= 0) epi.with_prefetched(r, acc[j], A.split, part, pre[j], sums);
        else if (rl >= cs_rr(A)) cs_slot_done(A, epi, part, c, rl, acc[j], sums, maxs, false);  // (a piece slot)
      }
    } else {
#pragma unroll
      for (int j = 0; j < RPT; ++j) cs_slot_done(A, epi, part, c, j * kCsThreads + tid, acc[j], sums, maxs, combine);
    }
  } else {
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    cs_slot_done(A, epi, part, c, j * kCsThreads + tid, acc[j], sums, maxs, combine);
  }
  }
  CS_TL_STAMP(4);
  CS_TL_FLUSH();
  if constexpr (Epi::kSums > 0 || Epi::kMaxs > 0) {
    // reduction partials: one slot per workgroup, or — combine mode — one per chunk (written by its last arriver)
    const int slot = A.pbase + (combine ? c : wg), nslot = A.pstride > 0 ? A.pstride : (combine ? A.nchunks : (int)gridDim.x);
#pragma unroll
    for (int i = 0; i < Epi::kSums; ++i) {
      const double t = block_sum<kCsThreads>(sums[i], red);
      if (tid == 0) epi.partial[(size_t)i * nslot + slot] = t;
    }
#pragma unroll
    for (int i = 0; i < Epi::kMaxs; ++i) {
      const double t = block_max<kCsThreads>(maxs[i], red);
      if (tid == 0) epi.partial[(size_t)(Epi::kSums + i) * nslot + slot] = t;
    }
  }
}

// 3 (default, round 4) = braided gathers / row sums with the stream loads of pass g + 2 issued BEFORE the barrier of step g
// (k_spmv_cs_il<.., 6>: -6..7 us per launch on the metric shapes, same bits; tools/cs_lab.hip, profiles/r04_cs_lab.txt);
// 2 = the braid of rounds 2-3 (stream loads behind the gathers); 1 = gather-ahead (k_spmv_cs_ga); 0 = k_spmv_cs
// The product launches schedule 3 only; the older schedules (and the in-kernel combine of split layouts, which lives in the stage-default
// instantiation of k_spmv_cs_il) are instantiated in the labs build (SCS_HIP_CS_SCHED, SCS_HIP_CS_COMBINE: options.hpp).
inline int cs_schedule() { return kLabsBuild ? opts().cs_sched : 3; }

template <class Epi>
inline void launch_spmv_cs(const CsView &A, const double *x, const Epi &epi, const int *done_flag, hipStream_t s,
                           int *step_counter) {
  if (A.nchunks <= 0) return;
  const dim3 g(A.nchunks * A.split), b(kCsThreads);
  if (cs_schedule() >= 3) {
    switch (A.rpt) {
      case 1: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 1, 6>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 2: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 2, 6>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 4: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 4, 6>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 8: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 8, 6>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      default: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 16, 6>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
    }
    return;
  }
#ifdef SCS_HIP_LABS
  if (cs_schedule() == 2 || (A.split > 1 && A.ticket != nullptr)) {  // (the in-kernel combine lives in k_spmv_cs_il only)
    switch (A.rpt) {
      case 1: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 1>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 2: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 2>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 4: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 4>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 8: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 8>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      default: hipLaunchKernelGGL((k_spmv_cs_il<Epi, 16>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
    }
    return;
  }
  if (cs_schedule() == 1) {
    switch (A.rpt) {
      case 1: hipLaunchKernelGGL((k_spmv_cs_ga<Epi, 1>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 2: hipLaunchKernelGGL((k_spmv_cs_ga<Epi, 2>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 4: hipLaunchKernelGGL((k_spmv_cs_ga<Epi, 4>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      case 8: hipLaunchKernelGGL((k_spmv_cs_ga<Epi, 8>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
      default: hipLaunchKernelGGL((k_spmv_cs_ga<Epi, 16>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
    }
    return;
  }
  switch (A.rpt) {
    case 1: hipLaunchKernelGGL((k_spmv_cs<Epi, 1>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
    case 2: hipLaunchKernelGGL((k_spmv_cs<Epi, 2>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
    case 4: hipLaunchKernelGGL((k_spmv_cs<Epi, 4>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
    case 8: hipLaunchKernelGGL((k_spmv_cs<Epi, 8>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
    default: hipLaunchKernelGGL((k_spmv_cs<Epi, 16>), g, b, 0, s, A, x, epi, done_flag, step_counter); break;
  }
#endif
}

}  // namespace scship
