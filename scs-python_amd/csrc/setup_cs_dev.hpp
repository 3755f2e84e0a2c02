// setup_cs_dev.hpp — the column-sorted pass layout of spmv_cs.hpp built on the device at scs_init.
//
// Input: T = CSR of the TRANSPOSE of the matrix M the layout is for (for M = A that is the caller's CSC(A) as it
// arrived, for M = A' the CSR(A) built by setup_dev.hpp, for the symmetric P itself).  T's nonzero array is M's
// nonzeros in (column, row) order, so the (column, row)-sorted stream of every row chunk of M is a STABLE
// partition of that array by chunk = row / R:
//   1. k_cs_hist     per block of 4096 consecutive nonzeros: how many fall into each chunk (LDS integer atomics);
//   2. exclusive scan over (chunk, block)  -> where each block's share of each chunk starts;
//   3. k_cs_scatter  per block: bitonic sort of (chunk << 12 | position) in LDS gives the stable rank inside the
//                    block; entries go to their place in the chunk streams {row, column, source position};
//   4. k_cs_cut      per chunk: greedy cut into passes (<= 8192 nonzeros, < 2^19 columns wide);
//   5. k_cs_fill     per pass: bitonic sort of (cs_row_key << 13 | position) = the LDS slot of every
//                    nonzero; writes val / idx in the lane-major quad order and the per-lane run descriptors.
// Integer work only; the result is identical, entry for entry, to spmv_cs.hpp build_cs (the host builder).
#pragma once
#include "common.hpp"
#include "setup_dev.hpp"
#include "spmv_cs.hpp"

namespace scship {

constexpr int kCsBlock = 4096;     // nonzeros per partition block
constexpr int kCsMaxChunks = 4096;

// ascending bitonic sort of N (power of two) keys in LDS by the whole workgroup
template <int N>
__device__ __forceinline__ void bitonic_sort_lds(unsigned *k) {
  for (int size = 2; size <= N; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < N / 2; t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned a = k[lo], b = k[hi];
        if ((a > b) == up) { k[lo] = b; k[hi] = a; }
      }
    }
  __syncthreads();
}

// hist[chunk * nblocks + block]
__global__ __launch_bounds__(1024) void k_cs_hist(const int *__restrict__ trow, long nnz, int R, int nchunks, int nblocks, int *hist,
                                                  const unsigned *__restrict__ peel) {
  __shared__ int cnt[kCsMaxChunks];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < nchunks; c += blockDim.x) cnt[c] = 0;
  __syncthreads();
  for (int l = threadIdx.x; l < kCsBlock; l += blockDim.x) {
    const long p = (long)b * kCsBlock + l;
    if (p < nnz && !cs_is_peeled(peel, trow[p])) atomicAdd(&cnt[trow[p] / R], 1);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < nchunks; c += blockDim.x) hist[(size_t)c * nblocks + b] = cnt[c];
}

__global__ __launch_bounds__(1024) void k_cs_scatter(const int *__restrict__ tptr, const int *__restrict__ trow, int trows, long nnz, int R,
                                                     int nchunks, int nblocks, const int *__restrict__ hoff, int *s_row, int *s_col,
                                                     int *s_src, const unsigned *__restrict__ peel) {
  __shared__ unsigned key[kCsBlock];
  __shared__ int first[kCsMaxChunks];
  const int b = blockIdx.x;
  for (int l = threadIdx.x; l < kCsBlock; l += blockDim.x) {
    const long p = (long)b * kCsBlock + l;
    key[l] = (p < nnz && !cs_is_peeled(peel, trow[p])) ? ((unsigned)(trow[p] / R) << 12) | (unsigned)l : 0xffffffffu;
  }
  bitonic_sort_lds<kCsBlock>(key);
  for (int i = threadIdx.x; i < kCsBlock; i += blockDim.x) {
    const unsigned k = key[i];
    if (k != 0xffffffffu && (i == 0 || (key[i - 1] >> 12) != (k >> 12))) first[k >> 12] = i;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCsBlock; i += blockDim.x) {
    const unsigned k = key[i];
    if (k == 0xffffffffu) continue;
    const int c = (int)(k >> 12);
    const long p = (long)b * kCsBlock + (k & 4095);
    const int dst = hoff[(size_t)c * nblocks + b] + (i - first[c]);
    // the row of T holding position p: largest j with tptr[j] <= p
    int lo = 0, hi = trows;  // invariant tptr[lo] <= p < tptr[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tptr[mid] <= p) lo = mid; else hi = mid;
    }
    s_row[dst] = trow[p];
    s_col[dst] = lo;
    s_src[dst] = (int)p;
  }
}

// Virtual rows (spmv_cs.hpp CsView::Rr): the row SLOT of every nonzero of T.  rowinfo[r] = {first piece, pieces} of a split row,
// {-1, 0} otherwise; the k-th nonzero (ascending column) of a split row goes to piece k mod np — k by binary search of the
// nonzero's column (the row of T holding position p) in M's own CSR row.
__global__ __launch_bounds__(256) void k_cs_vslot(const int *__restrict__ tptr, const int *__restrict__ trow, int trows, long nnz,
                                                  const int *__restrict__ mrowptr, const int *__restrict__ mcol,
                                                  const int2 *__restrict__ rowinfo, int Rr, int Rp, int R, int *__restrict__ vslot) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  const int r = trow[p];
  const int2 info = rowinfo[r];
  if (info.x < 0) { vslot[p] = cs_slot_of_row(r, Rr, R); return; }
  int lo = 0, hi = trows;  // the row of T (= column of M) holding position p: tptr[lo] <= p < tptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tptr[mid] <= p) lo = mid; else hi = mid;
  }
  const int c = lo;
  int a = mrowptr[r], b = mrowptr[r + 1];  // first position with mcol >= c
  while (a < b) {
    const int mid = (a + b) >> 1;
    if (mcol[mid] < c) a = mid + 1; else b = mid;
  }
  const int k = a - mrowptr[r];
  vslot[p] = cs_slot_of_piece(info.x + k % info.y, Rr, Rp, R);
}

// one workgroup: passes of every (chunk, part).  pass_info = {first stream position, nonzeros, first column, chunk}
__global__ __launch_bounds__(1024) void k_cs_cut(const int *__restrict__ hoff, int nblocks, int nchunks, int split, long nnz,
                                                 const int *__restrict__ s_col, int max_pass, int *passptr, int4 *pass_info, int *fail) {
  __shared__ int sm[kScanThreads / 64];
  __shared__ int carry_sm;
  auto cut = [&](int v, int4 *out) {  // passes of workgroup v = chunk * split + part (written from out[0] on when out != nullptr)
    const int c = v / split, part = v - c * split;
    const long c_begin = hoff[(size_t)c * nblocks], c_end = hoff[(size_t)(c + 1) * nblocks];  // (the scan's total closes the last chunk)
    // the chunk's stream cut into `split` parts at (multiples of 256 near) k * n / split
    const long e_begin = c_begin + cs_part_cut(c_end - c_begin, part, split), e_end = c_begin + cs_part_cut(c_end - c_begin, part + 1, split);
    int np = 0;
    long e0 = e_begin;
    const int plen = cs_pass_len(e_end - e_begin);
    while (e0 < e_end) {
      long e1 = e0 + plen < e_end ? e0 + plen : e_end;
      const int base = s_col[e0];
      if ((long)s_col[e1 - 1] - base >= (1L << kCsColBits)) {
        long lo = e0, hi = e1 - 1;  // s_col[lo] - base fits, s_col[hi] - base does not
        while (hi - lo > 1) {
          const long mid = (lo + hi) >> 1;
          if ((long)s_col[mid] - base >= (1L << kCsColBits)) hi = mid; else lo = mid;
        }
        e1 = hi;
      }
      if (out) out[np] = int4{(int)e0, (int)(e1 - e0), base, c};
      ++np;
      e0 = e1;
    }
    return np;
  };
  if (threadIdx.x == 0) carry_sm = 0;
  __syncthreads();
  const int nwg = nchunks * split;
  for (int v0 = 0; v0 < nwg; v0 += kScanThreads) {
    const int v = v0 + threadIdx.x;
    const int np = v < nwg ? cut(v, nullptr) : 0;
    int total;
    const int off = block_excl_scan(np, sm, total);
    const int start = carry_sm + off;
    if (v < nwg) {
      passptr[v] = start;
      if (start + np <= max_pass) cut(v, pass_info + start);
      else atomicExch(fail, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_sm += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) passptr[nwg] = carry_sm;
}

template <int RPT>
__global__ __launch_bounds__(1024) void k_cs_fill(const int4 *__restrict__ pass_info, const int *__restrict__ s_row,
                                                  const int *__restrict__ s_col, const int *__restrict__ s_src,
                                                  const double *__restrict__ tval, int R, unsigned *idx, double *val,
                                                  unsigned long long *meta, int2 *pinfo, int *fail) {
  constexpr int CB = RPT == 16 ? 6 : (48 / RPT < 13 ? 48 / RPT : 13);
  constexpr int RR = kCsThreads * RPT;
  __shared__ unsigned key[kCsPass];
  __shared__ int cnt[RR];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int4 pi = pass_info[g];
  const int begin = pi.x, len = pi.y, base = pi.z, r0 = pi.w * R;
  for (int k = tid; k < RR; k += kCsThreads) cnt[k] = 0;
  __syncthreads();
  for (int q = tid; q < kCsPass; q += kCsThreads) {
    unsigned k = (unsigned)RR;
    if (q < len) {
      const int rl = s_row[begin + q] - r0;
      k = (unsigned)cs_row_key(rl, RPT);
      atomicAdd(&cnt[k], 1);
    }
    key[q] = (k << kCsSlotBits) | (unsigned)q;
  }
  bitonic_sort_lds<kCsPass>(key);
  const size_t o = (size_t)g * kCsPass;
  for (int i = tid; i < kCsPass; i += kCsThreads) {
    const int q = (int)(key[i] & (kCsPass - 1)), sp = cs_store_pos(q);
    if (q < len) {
      const int e = begin + q;
      idx[o + sp] = ((unsigned)(s_col[e] - base) << kCsSlotBits) | (unsigned)i;
      val[o + sp] = tval[s_src[e]];
    } else {
      idx[o + sp] = (unsigned)i;
      val[o + sp] = 0.0;
    }
  }
  // run descriptor of lane tid: first slot of its run = entries with key < that of its first row
  const unsigned want = (unsigned)cs_row_key(tid, RPT) << kCsSlotBits;
  int lo = -1, hi = kCsPass;  // key[lo] < want <= key[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (key[mid] < want) lo = mid; else hi = mid;
  }
  unsigned long long w = (unsigned long long)hi, w1 = 0;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int n = cnt[cs_row_key(j * kCsThreads + tid, RPT)];
    if (n > cs_peel_threshold(RPT)) atomicExch(fail, 1);  // (build_cs: the field's capacity, and never more than 2048 per pass)
    const unsigned long long nb = (unsigned long long)(n & ((1 << CB) - 1));
    if (RPT < 16 || j < 8) w |= nb << (16 + CB * j);
    else w1 |= nb << (CB * (j - 8));
  }
  if (RPT == 16) {
    meta[((size_t)g * kCsThreads + tid) * 2] = w;
    meta[((size_t)g * kCsThreads + tid) * 2 + 1] = w1;
  } else {
    meta[(size_t)g * kCsThreads + tid] = w;
  }
  if (tid == 0) pinfo[g] = int2{base, len};
}

// Owning device copy of one matrix in the column-sorted pass layout
struct DeviceCs {
  DevBuf<int> passptr;
  DevBuf<int2> pinfo;
  DevBuf<unsigned> idx;
  DevBuf<double> val;
  DevBuf<unsigned long long> meta;
  DevBuf<double> scratch;    // in-kernel combine of split layouts (CsView::scratch / ticket); empty: partial outputs
  DevBuf<unsigned> ticket;
  int rows = 0, cols = 0, nchunks = 0, R = 0, npass = 0, rpt = 0, split = 1;
  int Rr = 0, Rp = 0, npieces = 0;  // virtual rows (CsView): real-row / piece slots per chunk, pieces in all
  DevBuf<double> tpart;             // their sums
  bool ok = false;
  void release() {
    passptr.release(); pinfo.release(); idx.release(); val.release(); meta.release(); scratch.release(); ticket.release(); tpart.release();
    Rr = Rp = npieces = 0;
    ok = false;
  }
  bool combine() const { return ok && split > 1 && ticket.p != nullptr; }
  // the workgroups of a chunk add their partial row sums inside the kernel (k_spmv_cs_il): every epilogue sees finished rows
  void enable_combine(hipStream_t s) {
    if (!ok || split <= 1) return;
    scratch.alloc_zero((size_t)nchunks * split * kCsThreads * rpt, s);
    ticket.alloc_zero((size_t)nchunks, s);
  }
  CsView view() const {
    CsView v{passptr.p, pinfo.p, idx.p, val.p, meta.p, rows, cols, nchunks, R, npass, rpt, split};
    v.scratch = scratch.p;
    v.ticket = ticket.p;
    v.Rr = Rr; v.Rp = Rp; v.npieces = npieces; v.tpart = tpart.p;
    return v;
  }
  void from_host(const HostCs &h, hipStream_t s) {
    Rr = Rp = npieces = 0;
    tpart.release();
    rows = h.rows; cols = h.cols; nchunks = h.nchunks; R = h.R; npass = h.npass; rpt = h.rpt; split = h.split;
    passptr.upload(h.passptr.data(), h.passptr.size(), s);
    pinfo.upload(h.pinfo.data(), h.pinfo.size(), s);
    idx.upload(h.idx.data(), h.idx.size(), s);
    val.upload(h.val.data(), h.val.size(), s);
    meta.upload(h.meta.data(), h.meta.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    ok = true;
  }
  // layout for the matrix M (rows_ x cols_) whose TRANSPOSE is the CSR (tptr, trow, tval) with cols_ rows.
  // false (and nothing kept) when the pattern does not fit the format: the caller keeps its other layouts.
  // force_R / force_rpt: the caller's geometry (virtual rows: `trow` holds row SLOTS, rows_ = nchunks * R of them)
  bool build_from_transpose(int rows_, int cols_, const int *tptr, const int *trow, const double *tval, long nnz, hipStream_t s,
                            int split_ = 1, const unsigned *peel = nullptr, int force_R = 0, int force_rpt = 0) {
    release();
    rows = rows_; cols = cols_; split = split_;
    cs_pick_geometry(rows, R, rpt, split);
    if (force_R > 0) { R = force_R; rpt = force_rpt; }
    nchunks = (rows + R - 1) / R;
    const int nblocks = (int)((nnz + kCsBlock - 1) / kCsBlock);
    if (nnz <= 0 || nchunks > kCsMaxChunks || (long)nchunks * nblocks > (long)kScanTile * kScanTile || nnz > 2000000000L) return false;
    const long nh = (long)nchunks * nblocks;
    DevBuf<int> hist, hoff, tmp, s_row, s_col, s_src, flag;
    DevBuf<int4> pass_info;
    hist.alloc((size_t)nh);
    hoff.alloc((size_t)nh + 1);
    tmp.alloc_zero((size_t)(nh / kScanTile + 4), s);
    flag.alloc_zero(1, s);
    s_row.alloc((size_t)nnz); s_col.alloc((size_t)nnz); s_src.alloc((size_t)nnz);
    hipLaunchKernelGGL(k_cs_hist, dim3(nblocks), dim3(1024), 0, s, trow, nnz, R, nchunks, nblocks, hist.p, peel);
    device_exclusive_scan(hist.p, hoff.p, nh, tmp.p, s);
    hipLaunchKernelGGL(k_cs_scatter, dim3(nblocks), dim3(1024), 0, s, tptr, trow, cols, nnz, R, nchunks, nblocks, hoff.p, s_row.p,
                       s_col.p, s_src.p, peel);
    // acceptance bound of build_cs: more passes than this means too much padding
    const long max_pass_l = (nnz + nnz / 4) / kCsPass + (long)nchunks * split + 1;
    const int max_pass = (int)std::min<long>(max_pass_l, 2000000000L / kCsPass);
    pass_info.alloc((size_t)max_pass);
    passptr.alloc((size_t)nchunks * split + 1);
    hipLaunchKernelGGL(k_cs_cut, dim3(1), dim3(kScanThreads), 0, s, hoff.p, nblocks, nchunks, split, nnz, s_col.p, max_pass, passptr.p,
                       pass_info.p, flag.p);
    int failed = 0;
    HIP_CHECK(hipMemcpyAsync(&npass, passptr.p + (size_t)nchunks * split, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&failed, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (failed || (long)npass * kCsPass > nnz + nnz / 4 + (long)nchunks * split * kCsPass) { release(); return false; }
    idx.alloc((size_t)npass * kCsPass);
    val.alloc((size_t)npass * kCsPass);
    meta.alloc((size_t)npass * kCsThreads * cs_meta_words(rpt));
    pinfo.alloc((size_t)npass);
    const dim3 gr(npass), bl(kCsThreads);
#define SCS_CS_FILL(RPT_) hipLaunchKernelGGL((k_cs_fill<RPT_>), gr, bl, 0, s, pass_info.p, s_row.p, s_col.p, s_src.p, tval, R, idx.p, val.p, meta.p, pinfo.p, flag.p)
    switch (rpt) {
      case 1: SCS_CS_FILL(1); break;
      case 2: SCS_CS_FILL(2); break;
      case 4: SCS_CS_FILL(4); break;
      case 8: SCS_CS_FILL(8); break;
      default: SCS_CS_FILL(16); break;
    }
#undef SCS_CS_FILL
    HIP_CHECK(hipMemcpyAsync(&failed, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (failed) { release(); return false; }
    ok = true;
    return true;
  }
};

}  // namespace scship
