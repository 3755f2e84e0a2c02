// setup_dev.hpp — scs_init's O(nnz) matrix work on the device: CSC -> CSR transposition and the construction of
// the L2-blocked slab layout (spmv.hpp), so that the host only validates and uploads the caller's CSC arrays.
//
// Plays the role of the matrix copies / transposition of scs_source/linsys/scs_matrix.c and the indirect
// backend's private.c init (R:meson.build:199-202,261; absent).  Measured at the bench size (nnz = 2e7): the host
// versions (host_setup.hpp csc_to_csr + build_slab x 2) were 690 of 820 ms of scs_init.
//
// Everything here is integer work with fixed results: histogram by integer atomics, exclusive scans, a scatter
// whose arbitrary within-row order is removed again by sorting every row on (column, source index).  The layouts
// produced are identical, entry for entry, to the host builders' (which remain as the fallback for matrices with
// rows too long for a one-lane sort, and as the reference in tests/test_hip_parity.py).
#pragma once
#include "common.hpp"
#include "spmv.hpp"
#include "vec.hpp"

namespace scship {

constexpr int kScanThreads = 1024;
constexpr int kScanPerThread = 4;
constexpr int kScanTile = kScanThreads * kScanPerThread;
constexpr int kSortMaxLen = 512;  // rows longer than this send the matrix down the host path
constexpr int kTransposeWaveRows = 32768;  // up to this many rows / columns: one wavefront per row in the transposition kernels

// ---- exclusive scan of int arrays (three launches; tile sums scanned by one workgroup: n <= 4096^2) ----
__device__ __forceinline__ int block_excl_scan(int v, int *sm, int &total) {  // kScanThreads lanes, result per lane
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) sm[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    int w = lane < kScanThreads / 64 ? sm[lane] : 0;
#pragma unroll
    for (int d = 1; d < kScanThreads / 64; d <<= 1) {
      const int t = __shfl_up(w, d, 64);
      if (lane >= d) w += t;
    }
    if (lane < kScanThreads / 64) sm[lane] = w;  // inclusive wave totals
  }
  __syncthreads();
  const int base = wid ? sm[wid - 1] : 0;
  total = sm[kScanThreads / 64 - 1];
  __syncthreads();
  return base + inc - v;
}
__global__ __launch_bounds__(kScanThreads) void k_scan_tiles(const int *in, int *out, long n, int *tile_sum) {
  __shared__ int sm[kScanThreads / 64];
  const long base = (long)blockIdx.x * kScanTile + (long)threadIdx.x * kScanPerThread;
  int v[kScanPerThread], s = 0;
#pragma unroll
  for (int i = 0; i < kScanPerThread; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
  int total;
  int off = block_excl_scan(s, sm, total);
#pragma unroll
  for (int i = 0; i < kScanPerThread; ++i) {
    if (base + i < n) out[base + i] = off;
    off += v[i];
  }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(int *tile_sum, int ntiles, int *grand_total) {
  __shared__ int sm[kScanThreads / 64];
  int carry = 0;
  for (int b0 = 0; b0 < ntiles; b0 += kScanThreads) {
    const int i = b0 + threadIdx.x;
    const int v = i < ntiles ? tile_sum[i] : 0;
    int total;
    const int off = block_excl_scan(v, sm, total);
    if (i < ntiles) tile_sum[i] = carry + off;
    carry += total;
  }
  if (threadIdx.x == 0) *grand_total = carry;
}
__global__ __launch_bounds__(kScanThreads) void k_scan_add(int *out, long n, const int *tile_off, const int *grand_total) {
  const long base = (long)blockIdx.x * kScanTile + (long)threadIdx.x * kScanPerThread;
  const int add = tile_off[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanPerThread; ++i)
    if (base + i < n) out[base + i] += add;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = *grand_total;  // closing offset
}
// out[0..n] = exclusive prefix sums of in[0..n)  (out has n + 1 entries; in == out allowed); tmp: ntiles + 1 ints
inline void device_exclusive_scan(const int *in, int *out, long n, int *tmp, hipStream_t s) {
  const int ntiles = (int)((n + kScanTile - 1) / kScanTile);
  if (ntiles > kScanTile) throw std::runtime_error("device_exclusive_scan: array too long");
  hipLaunchKernelGGL(k_scan_tiles, dim3(std::max(ntiles, 1)), dim3(kScanThreads), 0, s, in, out, n, tmp);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kScanThreads), 0, s, tmp, ntiles, tmp + ntiles);
  hipLaunchKernelGGL(k_scan_add, dim3(std::max(ntiles, 1)), dim3(kScanThreads), 0, s, out, n, tmp, tmp + ntiles);
}

// ---- CSC (= CSR of the transpose) -> CSR ----
__global__ __launch_bounds__(kVecThreads) void k_count_index(const int *__restrict__ idx, long nnz, int *cnt) {
  for (long p = (long)blockIdx.x * kVecThreads + threadIdx.x; p < nnz; p += (long)gridDim.x * kVecThreads) atomicAdd(&cnt[idx[p]], 1);
}
// entries of source row j (a column of the result) are appended to their rows in arbitrary order; src keeps p
__global__ __launch_bounds__(kVecThreads) void k_transpose_scatter(const int *__restrict__ sptr, const int *__restrict__ sidx, int srows,
                                                                   int *cursor, int *out_col, int *out_src) {
  for (long j = (long)blockIdx.x * kVecThreads + threadIdx.x; j < srows; j += (long)gridDim.x * kVecThreads)
    for (int p = sptr[j]; p < sptr[j + 1]; ++p) {
      const int q = atomicAdd(&cursor[sidx[p]], 1);
      out_col[q] = (int)j;
      out_src[q] = p;
    }
}
// one lane per row: insertion sort on (column, source index), then the values follow their source index
__global__ __launch_bounds__(kVecThreads) void k_sort_rows(const int *__restrict__ rowptr, int *col, int *src, int rows, int *too_long) {
  for (long r = (long)blockIdx.x * kVecThreads + threadIdx.x; r < rows; r += (long)gridDim.x * kVecThreads) {
    const int a = rowptr[r], e = rowptr[r + 1];
    if (e - a > kSortMaxLen) { atomicExch(too_long, 1); continue; }
    for (int i = a + 1; i < e; ++i) {
      const int c = col[i], s = src[i];
      int k = i - 1;
      while (k >= a && (col[k] > c || (col[k] == c && src[k] > s))) {
        col[k + 1] = col[k];
        src[k + 1] = src[k];
        --k;
      }
      col[k + 1] = c;
      src[k + 1] = s;
    }
  }
}
// Small matrices (round 5, late: the transposition of a config-5 member was two launches of ~100 and ~140 us — 40 dependent atomics per lane,
// an insertion sort in global memory per lane — in a chain of dependent dispatches whose length is what a batch's scs_init costs): ONE WAVEFRONT
// per source row / result row.  The result is the same: the scatter's order inside a row is arbitrary either way and the sort's order is
// total ((column, source index) is unique).
__global__ __launch_bounds__(kVecThreads) void k_transpose_scatter_w(const int *__restrict__ sptr, const int *__restrict__ sidx, int srows,
                                                                     int *cursor, int *out_col, int *out_src) {
  const int lane = threadIdx.x & 63;
  for (long j = (long)blockIdx.x * (kVecThreads / 64) + (threadIdx.x >> 6); j < srows; j += (long)gridDim.x * (kVecThreads / 64))
    for (int p = sptr[j] + lane; p < sptr[j + 1]; p += 64) {
      const int q = atomicAdd(&cursor[sidx[p]], 1);
      out_col[q] = (int)j;
      out_src[q] = p;
    }
}
// rank sort: lane i of the row's wavefront holds entry i and counts the entries that sort in front of it (rows up to 64 entries; longer
// ones: lane 0's insertion sort, as k_sort_rows)
__global__ __launch_bounds__(kVecThreads) void k_sort_rows_w(const int *__restrict__ rowptr, int *col, int *src, int rows, int *too_long) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * (kVecThreads / 64) + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * (kVecThreads / 64)) {
    const int a = rowptr[r], e = rowptr[r + 1], len = e - a;
    if (len > kSortMaxLen) {
      if (lane == 0) atomicExch(too_long, 1);
      continue;
    }
    if (len <= 64) {
      const int c = lane < len ? col[a + lane] : 0x7fffffff, sv = lane < len ? src[a + lane] : 0x7fffffff;
      int rank = 0;
      for (int k = 0; k < len; ++k) {
        const int ck = __shfl(c, k, 64), sk = __shfl(sv, k, 64);
        rank += (ck < c || (ck == c && sk < sv)) ? 1 : 0;
      }
      if (lane < len) {
        col[a + rank] = c;
        src[a + rank] = sv;
      }
    } else if (lane == 0) {
      for (int i = a + 1; i < e; ++i) {
        const int c = col[i], sv = src[i];
        int k = i - 1;
        while (k >= a && (col[k] > c || (col[k] == c && src[k] > sv))) {
          col[k + 1] = col[k];
          src[k + 1] = src[k];
          --k;
        }
        col[k + 1] = c;
        src[k + 1] = sv;
      }
    }
  }
}
__global__ __launch_bounds__(kVecThreads) void k_gather_f64(double *dst, const double *__restrict__ srcv, const int *__restrict__ idx, long n) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) dst[i] = srcv[idx[i]];
}

// ---- slab layout from a CSR matrix with ascending columns in every row ----
struct SlabGeom {
  int rows, cols, R, shift, S, nchunks;
};
// counts per (row, slab) into the padded row-offset array (as ushort counts; scanned in place afterwards)
__global__ __launch_bounds__(kVecThreads) void k_slab_count(const int *__restrict__ rowptr, const int *__restrict__ col, SlabGeom g,
                                                            unsigned short *roff, int *overflow) {
  for (long r = (long)blockIdx.x * kVecThreads + threadIdx.x; r < g.rows; r += (long)gridDim.x * kVecThreads) {
    const int c = (int)(r / g.R), rl = (int)(r - (long)c * g.R);
    int p = rowptr[r];
    const int e = rowptr[r + 1];
    while (p < e) {
      const int s = col[p] >> g.shift;
      int q = p + 1;
      while (q < e && (col[q] >> g.shift) == s) ++q;
      if (q - p > 65535) { atomicExch(overflow, 1); break; }
      roff[((size_t)c * g.S + s) * (g.R + kSlabRoffPad) + rl] = (unsigned short)(q - p);
      p = q;
    }
  }
}
// one workgroup per (chunk, slab) segment: counts -> exclusive offsets (R + 1 of them), padded size of the segment
__global__ __launch_bounds__(kScanThreads) void k_slab_scan(SlabGeom g, unsigned short *roff, int *seg_size, int *overflow) {
  __shared__ int sm[kScanThreads / 64];
  unsigned short *ro = roff + (size_t)blockIdx.x * (g.R + kSlabRoffPad);
  int carry = 0;
  for (int b0 = 0; b0 < g.R; b0 += kScanThreads) {
    const int i = b0 + threadIdx.x;
    const int v = i < g.R ? ro[i] : 0;
    int total;
    const int off = block_excl_scan(v, sm, total);
    if (i < g.R) ro[i] = (unsigned short)min(carry + off, 65535);
    carry += total;
  }
  if (threadIdx.x == 0) {
    if (carry > 65535) atomicExch(overflow, 1);
    ro[g.R] = (unsigned short)min(carry, 65535);
    seg_size[blockIdx.x] = (carry + 3) & ~3;  // segments are padded to a multiple of 4 entries
  }
}
__global__ __launch_bounds__(kVecThreads) void k_slab_fill(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                           const double *__restrict__ val, SlabGeom g, const unsigned short *__restrict__ roff,
                                                           const int *__restrict__ segptr, int *s_col, double *s_val) {
  for (long r = (long)blockIdx.x * kVecThreads + threadIdx.x; r < g.rows; r += (long)gridDim.x * kVecThreads) {
    const int c = (int)(r / g.R), rl = (int)(r - (long)c * g.R);
    int p = rowptr[r];
    const int e = rowptr[r + 1];
    while (p < e) {
      const int s = col[p] >> g.shift;
      const size_t seg = (size_t)c * g.S + s;
      int dst = segptr[seg] + roff[seg * (g.R + kSlabRoffPad) + rl];
      while (p < e && (col[p] >> g.shift) == s) {
        s_col[dst] = col[p];
        s_val[dst] = val[p];
        ++dst;
        ++p;
      }
    }
  }
}
// the up-to-3 padding entries at the end of every segment: zero values on a column inside the slab
__global__ __launch_bounds__(kVecThreads) void k_slab_pad(SlabGeom g, const unsigned short *__restrict__ roff, const int *__restrict__ segptr,
                                                          int *s_col, double *s_val) {
  const long nseg = (long)g.nchunks * g.S;
  for (long seg = (long)blockIdx.x * kVecThreads + threadIdx.x; seg < nseg; seg += (long)gridDim.x * kVecThreads) {
    const int s = (int)(seg % g.S);
    const int used = roff[(size_t)seg * (g.R + kSlabRoffPad) + g.R];
    const long padcol = min((long)s << g.shift, (long)g.cols - 1);
    for (int k = segptr[seg] + used; k < segptr[seg + 1]; ++k) {
      s_col[k] = (int)padcol;
      s_val[k] = 0.0;
    }
  }
}

}  // namespace scship
