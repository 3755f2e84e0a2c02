// slab_lab.hip — experiment bench for the L2-blocked SpMV (K1/K2) at the bench workload's shape:
// A is m x n with 20 nonzeros per column (=> Poisson(10) per row); runs the shipped kernel and
// experimental variants on CSR(A) (K1 shape) and CSR(A') (K2 shape), checks them bit-for-bit against
// a host CSR loop and prints the average launch time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o devtools/slab_lab tools/slab_lab.hip
//   ./devtools/slab_lab [m] [n] [nnz_per_col]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../scs-python_amd/csrc/spmv.hpp"
#include "../scs-python_amd/csrc/spmv_cs.hpp"

using namespace scship;
typedef int nt_i4 __attribute__((ext_vector_type(4)));
typedef double nt_d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ variant: gather-ahead pipeline
// Same slab format.  Differences from k_spmv_slab:
//  * rows are interleaved over the lanes (row = j*THREADS + tid): LDS runs of neighbouring lanes are
//    adjacent (no bank conflicts), offsets come as one unaligned dword {start,end} per row, the
//    epilogue's loads/stores are coalesced;
//  * the gathers of slab s+1 are issued BEFORE the LDS row sums of slab s, so the texture path is
//    busy while the wave sits in its LDS / barrier phase; values of s+1 and columns of s+2 are
//    streamed in the same window.
// ABL (ablations, results are wrong on purpose): 0 = none; 1 = no L2 gather (x index folded into 4 KB);
// 2 = values not streamed (1.0); 3 = neither values nor row offsets streamed; 4 = streaming loads nontemporal;
// 5 = values read from an L2-resident window
__device__ __forceinline__ int touch_lines(const void *base, long bytes, int lane, int nlanes) {
  const char *b = reinterpret_cast<const char *>(base);
  int acc = 0;
  for (long o0 = 0; o0 < bytes; o0 += 8L * nlanes * 128) {
    int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // all eight loads in flight before the first use
      const long off = o0 + ((long)lane + (long)k * nlanes) * 128;
      v[k] = off < bytes ? *reinterpret_cast<const int *>(b + off) : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
  }
  return acc;
}

// ------------------------------------------------------------------ experiment: scalar-cache prefetcher
// A separate tiny kernel (one wave per workgroup) on a second stream touches the val / col / roff lines of the
// slab format with SCALAR loads (s_load_dword: scalar data cache -> L2 -> HBM; not the TA/TCP path the compute
// waves gather through), in the order the compute workgroups consume them, so that their streaming loads hit
// L2 / Infinity Cache instead of waiting for HBM in the in-order vector-memory return queue.
__device__ __forceinline__ void s_touch16(const char *p) {  // 15 lines (128 B apart) in flight, then wait
  unsigned d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11, d12, d13, d14;
  asm volatile(
      "s_load_dword %0, %15, 0x0\n s_load_dword %1, %15, 0x80\n s_load_dword %2, %15, 0x100\n s_load_dword %3, %15, 0x180\n"
      "s_load_dword %4, %15, 0x200\n s_load_dword %5, %15, 0x280\n s_load_dword %6, %15, 0x300\n s_load_dword %7, %15, 0x380\n"
      "s_load_dword %8, %15, 0x400\n s_load_dword %9, %15, 0x480\n s_load_dword %10, %15, 0x500\n s_load_dword %11, %15, 0x580\n"
      "s_load_dword %12, %15, 0x600\n s_load_dword %13, %15, 0x680\n s_load_dword %14, %15, 0x700\n s_waitcnt lgkmcnt(0)"
      : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7), "=&s"(d8), "=&s"(d9),
        "=&s"(d10), "=&s"(d11), "=&s"(d12), "=&s"(d13), "=&s"(d14)
      : "s"(p)
      : "memory");
}
__device__ __forceinline__ void s_touch_range(const void *base, long bytes, int part, int nparts) {
  // lines [0, nl) of the range split into nparts contiguous shares; batches of 15 lines (tail over-reads < 2 KB:
  // callers pad their buffers)
  const char *b = reinterpret_cast<const char *>(base);
  const long nl = (bytes + 127) >> 7;
  const long per = (nl + nparts - 1) / nparts;
  long l0 = per * part, l1 = l0 + per < nl ? l0 + per : nl;
  for (long l = l0; l < l1; l += 15) {
    const char *p = b + (l << 7);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<unsigned long>(p) & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<unsigned long>(p) >> 32));
    s_touch16(reinterpret_cast<const char *>(((unsigned long)hi << 32) | lo));
  }
}

template <int THREADS, int RPT, int NQ, class Epi, int ABL = 0, int PF = 0, bool SPF = false>
__global__ __launch_bounds__(THREADS + 64 * PF) void k_slab_ga(SlabView A, const double *__restrict__ x, Epi epi) {
  constexpr int R = THREADS * RPT;
  constexpr int STAGE = 4 * NQ * THREADS;
  __shared__ __attribute__((aligned(16))) double prod[STAGE];
  const int tid = threadIdx.x, c = blockIdx.x;
  if (PF > 0 && tid >= THREADS) {
    // prefetch role: pull the lines the compute waves will stream at the top of the NEXT iteration into this
    // XCD's L2 (one dword per 128-byte line), so their in-order return queues never wait on HBM latency
    const int lane = tid - THREADS;
    const size_t sg = (size_t)c * A.S;
    int sink = 0;
    for (int s = -1; s < A.S; ++s) {
      int t0 = 0, t1 = 0, t2 = 0;
      if (SPF) {
        const int part = lane >> 6;
        if (s + 2 < A.S) {
          const int q0 = A.segptr[sg + s + 2], q1 = A.segptr[sg + s + 3];
          s_touch_range(A.val + q0, (long)(q1 - q0) * 8, part, PF);
          s_touch_range(A.roff + (sg + s + 2) * (R + kSlabRoffPad), (long)(R + 1) * 2, part, PF);
        }
        if (s + 3 < A.S) {
          const int q0 = A.segptr[sg + s + 3], q1 = A.segptr[sg + s + 4];
          s_touch_range(A.col + q0, (long)(q1 - q0) * 4, part, PF);
        }
      } else if (s + 2 < A.S) {
        const int q0 = A.segptr[sg + s + 2], q1 = A.segptr[sg + s + 3];
        t0 = touch_lines(A.val + q0, (long)(q1 - q0) * 8, lane, 64 * PF);
        t1 = touch_lines(A.roff + (sg + s + 2) * (R + kSlabRoffPad), (long)(R + 1) * 2, lane, 64 * PF);
      }
      if (!SPF && s + 3 < A.S) {
        const int q0 = A.segptr[sg + s + 3], q1 = A.segptr[sg + s + 4];
        t2 = touch_lines(A.col + q0, (long)(q1 - q0) * 4, lane, 64 * PF);
      }
      sink ^= t0 ^ t1 ^ t2;
      __syncthreads();
      if (s >= 0) __syncthreads();
    }
    if (sink == 0x5a5a5a5a && A.rows < 0) prod[0] = 1.;
    return;
  }
  double acc[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) acc[j] = 0.;
  const size_t seg0 = (size_t)c * A.S;
  int4 cc[NQ];
  double2 va[NQ], vb[NQ];
  double xg[NQ][4];
  unsigned o_cur[RPT], o_nxt[RPT];

  auto load_cols = [&](int p, int cnt4) {
    const int4 *c4 = reinterpret_cast<const int4 *>(A.col + p);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * THREADS;
      if (q < cnt4) {
        if (ABL == 4) { nt_i4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_i4 *>(&c4[q])); cc[i] = int4{t.x, t.y, t.z, t.w}; }
        else cc[i] = c4[q];
        if (ABL == 1) { cc[i].x &= 511; cc[i].y &= 511; cc[i].z &= 511; cc[i].w &= 511; }
      }
    }
  };
  auto load_vals = [&](int p, int cnt4) {
    const double2 *v2 = reinterpret_cast<const double2 *>(A.val + p);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * THREADS;
      if (q < cnt4) {
        if (ABL == 2 || ABL == 3) { va[i] = double2{1., 1.}; vb[i] = double2{1., 1.}; }
        else if (ABL == 4) {
          nt_d2 t0 = __builtin_nontemporal_load(reinterpret_cast<const nt_d2 *>(&v2[2 * q]));
          nt_d2 t1 = __builtin_nontemporal_load(reinterpret_cast<const nt_d2 *>(&v2[2 * q + 1]));
          va[i] = double2{t0.x, t0.y}; vb[i] = double2{t1.x, t1.y};
        }
        else if (ABL == 5) {  // values come from a 1 MB window (L2 hits): same TA/TCP work, no HBM latency
          const double2 *w2 = reinterpret_cast<const double2 *>(A.val + (p & 0x1ffff & ~3));
          va[i] = w2[2 * q]; vb[i] = w2[2 * q + 1];
        }
        else { va[i] = v2[2 * q]; vb[i] = v2[2 * q + 1]; }
      }
    }
  };
  auto gather = [&](int cnt4) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * THREADS;
      if (q < cnt4) { xg[i][0] = x[cc[i].x]; xg[i][1] = x[cc[i].y]; xg[i][2] = x[cc[i].z]; xg[i][3] = x[cc[i].w]; }
    }
  };
  auto load_offs = [&](size_t seg, unsigned *o) {
    const unsigned short *ro = A.roff + seg * (R + kSlabRoffPad) + tid;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      unsigned u;
      if (ABL == 3) u = ((unsigned)(tid + 1) << 16) | (unsigned)tid;
      else __builtin_memcpy(&u, ro + j * THREADS, 4);  // {start, end} of row j*THREADS + tid
      o[j] = u;
    }
  };
  auto stage = [&](int cnt4) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * THREADS;
      if (q < cnt4) {
        double2 a, b;
        a.x = va[i].x * xg[i][0]; a.y = va[i].y * xg[i][1];
        b.x = vb[i].x * xg[i][2]; b.y = vb[i].y * xg[i][3];
        reinterpret_cast<double2 *>(prod)[2 * q] = a;
        reinterpret_cast<double2 *>(prod)[2 * q + 1] = b;
      }
    }
  };

  // prologue: slab 0 -> LDS, columns of slab 1 in registers
  int pa = A.segptr[seg0], pb = A.segptr[seg0 + 1];
  int cnt4 = (pb - pa) >> 2;
  load_cols(pa, cnt4);
  load_vals(pa, cnt4);
  load_offs(seg0, o_cur);
  gather(cnt4);
  int cnt4_n = 0, pn = pb;
  stage(cnt4);
  if (A.S > 1) {
    const int pe = A.segptr[seg0 + 2];
    cnt4_n = (pe - pb) >> 2;
    load_cols(pb, cnt4_n);
    pn = pb;
    pb = pe;
  }
  __syncthreads();

  for (int s = 0; s < A.S; ++s) {
    const bool more = s + 1 < A.S;
    int cnt4_nn = 0;
    if (more) {
      gather(cnt4_n);        // slab s+1: in flight during the row sums below
      load_vals(pn, cnt4_n);
      load_offs(seg0 + s + 1, o_nxt);
      if (s + 2 < A.S) {
        const int pe = A.segptr[seg0 + s + 3];
        cnt4_nn = (pe - pb) >> 2;
        load_cols(pb, cnt4_nn);  // cc is free again: the gathers above have been issued
        pn = pb;
        pb = pe;
      }
    }
    // row sums of slab s from LDS (ascending column order inside every row)
#pragma unroll
    for (int j0 = 0; j0 < RPT; j0 += 4) {
      double v[4][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int a = o_cur[j0 + jj] & 0xffff;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[jj][i] = prod[min(a + i, STAGE - 1)];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int a = o_cur[j0 + jj] & 0xffff, e = o_cur[j0 + jj] >> 16;
        double t = acc[j0 + jj];
#pragma unroll
        for (int i = 0; i < 4; ++i) t = (a + i < e) ? t + v[jj][i] : t;
        for (int k = a + 4; k < e; ++k) t += prod[k];
        acc[j0 + jj] = t;
      }
    }
    __syncthreads();
    if (more) stage(cnt4_n);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; ++j) o_cur[j] = o_nxt[j];
    cnt4_n = cnt4_nn;
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int r = c * R + j * THREADS + tid;
    if (r < A.rows) epi(r, acc[j], nullptr, nullptr);
  }
}


// workgroup w (one wave): XCD x = w % 8, k = w / 8; chunk c = 8 * (k / WPC) + x, share k % WPC of every segment of c
__global__ __launch_bounds__(64) void k_pf_scalar(SlabView A, int WPC, int what) {
  const int w = blockIdx.x, xcd = w & 7, k = w >> 3;
  const int c = 8 * (k / WPC) + xcd, part = k % WPC;
  if (c >= A.nchunks) return;
  const size_t sg = (size_t)c * A.S;
  for (int s = 0; s < A.S; ++s) {
    const int q0 = A.segptr[sg + s], q1 = A.segptr[sg + s + 1];
    if (what & 1) s_touch_range(A.val + q0, (long)(q1 - q0) * 8, part, WPC);
    if (what & 2) s_touch_range(A.col + q0, (long)(q1 - q0) * 4, part, WPC);
    if (what & 4) s_touch_range(A.roff + (sg + s) * (A.R + kSlabRoffPad), (long)(A.R + 1) * 2, part, WPC);
  }
}

// ------------------------------------------------------------------ host side
struct Csr {
  int rows, cols;
  std::vector<int> rowptr, col;
  std::vector<double> val;
};

static void transpose(const Csr &a, Csr &t) {
  t.rows = a.cols; t.cols = a.rows;
  const long nnz = a.rowptr[a.rows];
  t.rowptr.assign(t.rows + 1, 0);
  t.col.resize(nnz); t.val.resize(nnz);
  for (long p = 0; p < nnz; ++p) t.rowptr[a.col[p] + 1]++;
  for (int i = 0; i < t.rows; ++i) t.rowptr[i + 1] += t.rowptr[i];
  std::vector<int> next(t.rowptr.begin(), t.rowptr.end() - 1);
  for (int r = 0; r < a.rows; ++r)
    for (int p = a.rowptr[r]; p < a.rowptr[r + 1]; ++p) {
      const int q = next[a.col[p]]++;
      t.col[q] = r; t.val[q] = a.val[p];
    }
}

template <class T>
static T *to_dev(const std::vector<T> &h) {
  T *d;
  HIP_CHECK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
  HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

struct DevSlab {
  SlabView v{};
  int *segptr, *col; unsigned short *roff; double *val;
};
static DevSlab upload_slab(const HostSlab &h) {
  DevSlab d;
  d.segptr = to_dev(h.segptr); d.col = to_dev(h.col); d.roff = to_dev(h.roff); d.val = to_dev(h.val);
  d.v = SlabView{d.segptr, d.roff, d.col, d.val, h.rows, h.cols, h.nchunks, h.S, h.R, h.max_seg};
  return d;
}
static void free_slab(DevSlab &d) { hipFree(d.segptr); hipFree(d.col); hipFree(d.roff); hipFree(d.val); }

template <class F>
static double time_us(F launch, int reps) {
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  HIP_CHECK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  HIP_CHECK(hipEventRecord(b)); HIP_CHECK(hipEventSynchronize(b));
  float ms; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / reps;
}

static long mismatches(const double *dy, const std::vector<double> &ref) {
  std::vector<double> h(ref.size());
  HIP_CHECK(hipMemcpy(h.data(), dy, ref.size() * sizeof(double), hipMemcpyDeviceToHost));
  long bad = 0;
  for (size_t i = 0; i < ref.size(); ++i) bad += std::memcmp(&h[i], &ref[i], 8) != 0;
  return bad;
}

template <int THREADS, int RPT, int NQ, int ABL = 0, int PF = 0, bool SPF = false>
static void run_ga(const char *tag, const Csr &M, const double *dx, double *dy, const std::vector<double> &ref) {
  HostSlab hs;
  if (!build_slab(M.rowptr.data(), M.col.data(), M.val.data(), M.rows, M.cols, hs, nullptr, THREADS * RPT)) { std::printf("  %-34s build failed\n", tag); return; }
  if (hs.R != THREADS * RPT) { std::printf("  %-34s R mismatch %d\n", tag, hs.R); return; }
  if (hs.max_seg > 4 * NQ * THREADS) { std::printf("  %-34s max_seg %d > stage %d\n", tag, hs.max_seg, 4 * NQ * THREADS); return; }
  for (int i = 0; i < 1024; ++i) { hs.col.push_back(0); hs.val.push_back(0.); hs.roff.push_back(0); }  // over-read pad; the dword read of the very last row's {start,end} pair stays in bounds
  DevSlab d = upload_slab(hs);
  HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
  auto launch = [&] { hipLaunchKernelGGL((k_slab_ga<THREADS, RPT, NQ, EpiStore, ABL, PF, SPF>), dim3(hs.nchunks), dim3(THREADS + 64 * PF), 0, 0, d.v, dx, EpiStore{dy, 0}); };
  const double us = time_us(launch, 20);
  std::printf("  %-34s R=%5d S=%2d wgs=%4d max_seg=%5d : %7.1f us  mismatches %ld\n", tag, hs.R, hs.S, hs.nchunks, hs.max_seg, us,
              mismatches(dy, ref));
  free_slab(d);
}

static void run_base(const char *tag, int rpt, const Csr &M, const double *dx, double *dy, const std::vector<double> &ref) {
  HostSlab hs;
  const bool ok = build_slab(M.rowptr.data(), M.col.data(), M.val.data(), M.rows, M.cols, hs, nullptr, 256 * rpt);
  if (!ok) { std::printf("  %-34s build failed\n", tag); return; }
  DevSlab d = upload_slab(hs);
  SpmvMat mat; mat.slab = d.v; mat.use_slab = true;
  HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
  auto launch = [&] { launch_spmv(mat, dx, EpiStore{dy, 0}, nullptr, 0); };
  const double us = time_us(launch, 20);
  std::printf("  %-34s R=%5d S=%2d wgs=%4d max_seg=%5d : %7.1f us  mismatches %ld\n", tag, hs.R, hs.S, hs.nchunks, hs.max_seg, us,
              mismatches(dy, ref));
  free_slab(d);
}


static void run_base_pf(const char *tag, int rpt, int wpc, int what, const Csr &M, const double *dx, double *dy,
                        const std::vector<double> &ref) {
  HostSlab hs;
  const bool ok = build_slab(M.rowptr.data(), M.col.data(), M.val.data(), M.rows, M.cols, hs, nullptr, 256 * rpt);
  if (!ok) { std::printf("  %-34s build failed\n", tag); return; }
  for (int i = 0; i < 1024; ++i) { hs.col.push_back(0); hs.val.push_back(0.); hs.roff.push_back(0); }  // over-read pad
  DevSlab d = upload_slab(hs);
  SpmvMat mat; mat.slab = d.v; mat.use_slab = true;
  HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
  hipStream_t s1, s2;
  HIP_CHECK(hipStreamCreate(&s1)); HIP_CHECK(hipStreamCreate(&s2));
  const int reps = 20, npf = ((hs.nchunks + 7) / 8) * 8 * wpc;
  std::vector<hipEvent_t> ev(reps + 4);
  for (auto &e : ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
  auto one = [&](int i) {
    if (i > 0) HIP_CHECK(hipStreamWaitEvent(s2, ev[i - 1], 0));
    hipLaunchKernelGGL(k_pf_scalar, dim3(npf), dim3(64), 0, s2, d.v, wpc, what);
    launch_spmv(mat, dx, EpiStore{dy, 0}, nullptr, s1);
    HIP_CHECK(hipEventRecord(ev[i], s1));
  };
  for (int i = 0; i < 3; ++i) one(i);
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipEventRecord(a, s1));
  for (int i = 0; i < reps; ++i) one(i);
  HIP_CHECK(hipEventRecord(b, s1));
  HIP_CHECK(hipDeviceSynchronize());
  float ms; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  // the prefetcher alone
  const double pf_us = time_us([&] { hipLaunchKernelGGL(k_pf_scalar, dim3(npf), dim3(64), 0, 0, d.v, wpc, what); }, 10);
  std::printf("  %-34s R=%5d S=%2d wgs=%4d pf_wgs=%5d : %7.1f us  (prefetcher alone %7.1f us)  mismatches %ld\n", tag, hs.R, hs.S,
              hs.nchunks, npf, ms * 1e3 / reps, pf_us, mismatches(dy, ref));
  HIP_CHECK(hipStreamDestroy(s1)); HIP_CHECK(hipStreamDestroy(s2));
  free_slab(d);
}

template <int RPT, int ABL>
static void run_cs_abl(const char *tag, const Csr &M, const double *dx, double *dy) {
  HostCs hc;
  if (!build_cs(M.rowptr.data(), M.col.data(), M.val.data(), M.rows, M.cols, hc, RPT)) { std::printf("  %-34s build failed\n", tag); return; }
  int *passptr = to_dev(hc.passptr);
  int2 *pinfo = to_dev(hc.pinfo);
  unsigned *idx = to_dev(hc.idx);
  double *val = to_dev(hc.val);
  unsigned long long *meta = to_dev(hc.meta);
  CsView v{passptr, pinfo, idx, val, meta, hc.rows, hc.cols, hc.nchunks, hc.R, hc.npass, hc.rpt, hc.split};
  auto launch = [&] { hipLaunchKernelGGL((k_spmv_cs_ga<EpiStore, RPT, ABL>), dim3(hc.nchunks), dim3(kCsThreads), 0, 0, v, dx, EpiStore{dy, 0}, nullptr, nullptr); };
  const double us = time_us(launch, 20);
  std::printf("  %-44s R=%5d : %7.1f us\n", tag, hc.R, us);
  hipFree(passptr); hipFree(pinfo); hipFree(idx); hipFree(val); hipFree(meta);
}

static void run_cs(const char *tag, int rpt, const Csr &M, const double *dx, double *dy, const std::vector<double> &ref) {
  HostCs hc;
  if (!build_cs(M.rowptr.data(), M.col.data(), M.val.data(), M.rows, M.cols, hc, rpt)) { std::printf("  %-34s build failed\n", tag); return; }
  int *passptr = to_dev(hc.passptr);
  int2 *pinfo = to_dev(hc.pinfo);
  unsigned *idx = to_dev(hc.idx);
  double *val = to_dev(hc.val);
  unsigned long long *meta = to_dev(hc.meta);
  CsView v{passptr, pinfo, idx, val, meta, hc.rows, hc.cols, hc.nchunks, hc.R, hc.npass, hc.rpt, hc.split};
  HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
  auto launch = [&] { launch_spmv_cs(v, dx, EpiStore{dy, 0}, nullptr, 0, nullptr); };
  const double us = time_us(launch, 20);
  std::printf("  %-34s R=%5d wgs=%4d passes=%5d (%.1f%% padding) : %7.1f us  mismatches %ld\n", tag, hc.R, hc.nchunks, hc.npass,
              100. * ((double)hc.npass * kCsPass / M.rowptr[M.rows] - 1.), us, mismatches(dy, ref));
  hipFree(passptr); hipFree(pinfo); hipFree(idx); hipFree(val); hipFree(meta);
}

struct EpiRaw2 {  // lab: raw partial row sums of the two halves
  double *y0, *y1;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { y0[r] = s; }
  __device__ void split(int r, double s, int part, double *, double *) const { (part ? y1 : y0)[r] = s; }
};
static void run_cs_split(const char *tag, const Csr &M, const double *dx, double *dy, const std::vector<double> &ref) {
  HostCs hc;
  if (!build_cs(M.rowptr.data(), M.col.data(), M.val.data(), M.rows, M.cols, hc, 0, 2)) { std::printf("  %-34s build failed\n", tag); return; }
  int *passptr = to_dev(hc.passptr);
  int2 *pinfo = to_dev(hc.pinfo);
  unsigned *idx = to_dev(hc.idx);
  double *val = to_dev(hc.val);
  unsigned long long *meta = to_dev(hc.meta);
  double *dy1;
  HIP_CHECK(hipMalloc(&dy1, ref.size() * 8));
  CsView v{passptr, pinfo, idx, val, meta, hc.rows, hc.cols, hc.nchunks, hc.R, hc.npass, hc.rpt, hc.split};
  auto launch = [&] { launch_spmv_cs(v, dx, EpiRaw2{dy, dy1}, nullptr, 0, nullptr); };
  const double us = time_us(launch, 20);
  std::vector<double> h0(ref.size()), h1(ref.size());
  HIP_CHECK(hipMemcpy(h0.data(), dy, ref.size() * 8, hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(h1.data(), dy1, ref.size() * 8, hipMemcpyDeviceToHost));
  double err = 0, scl = 0;
  for (size_t i = 0; i < ref.size(); ++i) { err = std::max(err, std::fabs(h0[i] + h1[i] - ref[i])); scl = std::max(scl, std::fabs(ref[i])); }
  std::printf("  %-34s R=%5d rpt=%2d wgs=%4d passes=%5d : %7.1f us  max err %.2e (scale %.1f)\n", tag, hc.R, hc.rpt, hc.nchunks * hc.split,
              hc.npass, us, err, scl);
  hipFree(passptr); hipFree(pinfo); hipFree(idx); hipFree(val); hipFree(meta); hipFree(dy1);
}

static void bench_matrix(const char *name, const Csr &M) {
  std::printf("%s: %d x %d, nnz %d\n", name, M.rows, M.cols, M.rowptr[M.rows]);
  std::vector<double> x(M.cols), ref(M.rows);
  std::mt19937_64 g(7);
  std::normal_distribution<double> nd;
  for (auto &v : x) v = nd(g);
  for (int r = 0; r < M.rows; ++r) {
    double s = 0.;
    for (int p = M.rowptr[r]; p < M.rowptr[r + 1]; ++p) s += M.val[p] * x[M.col[p]];
    ref[r] = s;
  }
  double *dx = to_dev(x), *dy;
  HIP_CHECK(hipMalloc(&dy, M.rows * sizeof(double)));
  run_base("shipped k_spmv_slab rpt16", 16, M, dx, dy, ref);
  run_base("shipped k_spmv_slab rpt8", 8, M, dx, dy, ref);
  if (getenv("LAB_CS_ABL")) {
    if (M.rows > M.cols) {
      run_cs_abl<8, 0>("cs gather-ahead rpt8", M, dx, dy);
      run_cs_abl<8, 1>("  abl1: gathers from a 2 KB table", M, dx, dy);
      run_cs_abl<8, 2>("  abl2: no LDS row sums", M, dx, dy);
      run_cs_abl<8, 3>("  abl3: no LDS traffic", M, dx, dy);
      run_cs_abl<8, 4>("  abl4: values not streamed", M, dx, dy);
    } else {
      run_cs_abl<4, 0>("cs gather-ahead rpt4", M, dx, dy);
      run_cs_abl<4, 1>("  abl1: gathers from a 2 KB table", M, dx, dy);
      run_cs_abl<4, 2>("  abl2: no LDS row sums", M, dx, dy);
      run_cs_abl<4, 3>("  abl3: no LDS traffic", M, dx, dy);
      run_cs_abl<4, 4>("  abl4: values not streamed", M, dx, dy);
    }
    hipFree(dx); hipFree(dy);
    return;
  }
  if (getenv("LAB_CS")) {
    run_cs_split("column-sorted, 2 workgroups/chunk", M, dx, dy, ref);
    for (int rpt : {0}) run_cs(cs_schedule() ? "column-sorted passes, gather-ahead" : "column-sorted passes", rpt, M, dx, dy, ref);
    hipFree(dx); hipFree(dy);
    return;
  }
  if (getenv("LAB_SPF")) {
    run_ga<256, 8, 3>("gather-ahead 256x8 nq3", M, dx, dy, ref);
    run_ga<256, 8, 3, 0, 2, true>("  + 2 scalar pf waves", M, dx, dy, ref);
    run_ga<256, 8, 3, 0, 4, true>("  + 4 scalar pf waves", M, dx, dy, ref);
    run_ga<512, 4, 2>("gather-ahead 512x4 nq2", M, dx, dy, ref);
    run_ga<512, 4, 2, 0, 2, true>("  + 2 scalar pf waves", M, dx, dy, ref);
    run_ga<512, 4, 2, 0, 4, true>("  + 4 scalar pf waves", M, dx, dy, ref);
    run_ga<512, 4, 2, 0, 8, true>("  + 8 scalar pf waves", M, dx, dy, ref);
    run_ga<256, 16, 6>("gather-ahead 256x16 nq6", M, dx, dy, ref);
    run_ga<256, 16, 6, 0, 2, true>("  + 2 scalar pf waves", M, dx, dy, ref);
    hipFree(dx); hipFree(dy);
    return;
  }
  if (getenv("LAB_PF")) {
    for (int rpt : {16, 8}) {
      run_base_pf("shipped + scalar pf x2 val+col", rpt, 2, 3, M, dx, dy, ref);
      run_base_pf("shipped + scalar pf x4 val+col", rpt, 4, 3, M, dx, dy, ref);
      run_base_pf("shipped + scalar pf x4 all", rpt, 4, 7, M, dx, dy, ref);
      run_base_pf("shipped + scalar pf x8 all", rpt, 8, 7, M, dx, dy, ref);
      run_base_pf("shipped + scalar pf x4 val", rpt, 4, 1, M, dx, dy, ref);
    }
    hipFree(dx); hipFree(dy);
    return;
  }
  run_ga<256, 16, 6>("gather-ahead 256x16 nq6", M, dx, dy, ref);
  run_ga<256, 8, 3>("gather-ahead 256x8 nq3", M, dx, dy, ref);
  run_ga<512, 4, 2>("gather-ahead 512x4 nq2", M, dx, dy, ref);
  run_ga<256, 16, 6, 5>("  abl5 values from L2 256x16", M, dx, dy, ref);
  run_ga<512, 4, 2, 1>("  abl1 no L2 gather 512x4", M, dx, dy, ref);
  run_ga<512, 4, 2, 2>("  abl2 no values 512x4", M, dx, dy, ref);
  run_ga<512, 4, 2, 3>("  abl3 no values/roff 512x4", M, dx, dy, ref);
  run_ga<512, 4, 2, 5>("  abl5 values from L2 512x4", M, dx, dy, ref);
  run_ga<512, 4, 2, 0, 1>("prefetch wave 512x4 +1", M, dx, dy, ref);
  run_ga<512, 4, 2, 0, 2>("prefetch wave 512x4 +2", M, dx, dy, ref);
  run_ga<256, 8, 3, 0, 1>("prefetch wave 256x8 +1", M, dx, dy, ref);
  run_ga<256, 16, 6, 0, 1>("prefetch wave 256x16 +1", M, dx, dy, ref);
  hipFree(dx); hipFree(dy);
}

int main(int argc, char **argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 2000000, n = argc > 2 ? atoi(argv[2]) : 1000000;
  const int k = argc > 3 ? atoi(argv[3]) : 20;
  // CSR(A') = the caller's CSC(A): n rows, k random distinct-ish columns in [0, m) per row, ascending
  Csr At;
  At.rows = n; At.cols = m;
  At.rowptr.resize(n + 1);
  At.col.reserve((size_t)n * k); At.val.reserve((size_t)n * k);
  std::mt19937_64 g(5);
  std::normal_distribution<double> nd;
  std::vector<int> tmp(k);
  At.rowptr[0] = 0;
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < k; ++i) tmp[i] = (int)(g() % (unsigned long)m);
    std::sort(tmp.begin(), tmp.end());
    int last = -1;
    for (int i = 0; i < k; ++i)
      if (tmp[i] != last) { At.col.push_back(tmp[i]); At.val.push_back(nd(g)); last = tmp[i]; }
    At.rowptr[j + 1] = (int)At.col.size();
  }
  Csr Ar;
  transpose(At, Ar);
  bench_matrix("K1 shape  CSR(A)", Ar);
  bench_matrix("K2 shape  CSR(A')", At);
  return 0;
}
