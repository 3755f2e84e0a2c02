// cs_lab.hip — timeline + variant bench of the column-sorted pass SpMV (spmv_cs.hpp) at the bench workload's shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/cs_lab tools/cs_lab.hip && ./gpurun_out/cs_lab [m] [n] [nnz_per_col]
// Prints, for the K1 (CSR(A)) and K2 (CSR(A'), split layout) shapes: the shipped kernel's time and a per-phase cycle
// breakdown (s_memtime stamps inside an instrumented copy of k_spmv_cs_ga): wait-for-gathers + product scatter,
// barrier wait, issue of the next gathers / stream loads, LDS row sums, prologue, epilogue.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <hip/hip_runtime.h>
#define CS_LAB_TIMELINE 1
__device__ unsigned long long cs_lab_tl[256 * 16 * 8];
#include "../scs-python_amd/csrc/spmv.hpp"
#include "../scs-python_amd/csrc/spmv_cs.hpp"

using namespace scship;
namespace scship { void set_last_error(const std::string &) {} }

struct Csr { int rows = 0, cols = 0; std::vector<int> rowptr, col; std::vector<double> val; };
static void transpose(const Csr &A, Csr &T) {
  T.rows = A.cols; T.cols = A.rows;
  T.rowptr.assign(T.rows + 1, 0);
  const int nnz = A.rowptr[A.rows];
  for (int p = 0; p < nnz; ++p) T.rowptr[A.col[p] + 1]++;
  for (int r = 0; r < T.rows; ++r) T.rowptr[r + 1] += T.rowptr[r];
  T.col.resize(nnz); T.val.resize(nnz);
  std::vector<int> cur(T.rowptr.begin(), T.rowptr.end() - 1);
  for (int r = 0; r < A.rows; ++r)
    for (int p = A.rowptr[r]; p < A.rowptr[r + 1]; ++p) { const int q = cur[A.col[p]]++; T.col[q] = r; T.val[q] = A.val[p]; }
}
template <class T> static T *to_dev(const std::vector<T> &h) {
  T *d; HIP_CHECK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
  HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
template <class F> static double time_us(F f, int reps) {
  for (int i = 0; i < 3; ++i) f();
  hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
  HIP_CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  HIP_CHECK(hipEventRecord(b, 0)); HIP_CHECK(hipEventSynchronize(b));
  float ms; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / reps;
}
static long mismatches(const double *dy, const std::vector<double> &ref) {
  std::vector<double> h(ref.size());
  HIP_CHECK(hipMemcpy(h.data(), dy, ref.size() * 8, hipMemcpyDeviceToHost));
  long bad = 0;
  for (size_t i = 0; i < ref.size(); ++i) bad += std::memcmp(&h[i], &ref[i], 8) != 0;
  return bad;
}

// ---- instrumented copy of k_spmv_cs_ga (same schedule) ----
enum { T_PRO = 0, T_WAIT_SCATTER, T_BARRIER, T_ISSUE, T_ROWSUM, T_EPI, T_TOTAL, T_N };
template <class Epi, int RPT>
__global__ __launch_bounds__(kCsThreads) void k_cs_timed(CsView A, const double *__restrict__ x, Epi epi, unsigned long long *tl) {
  constexpr int NQ = kCsQuads;
  __shared__ __attribute__((aligned(16))) double prod[2][kCsPass];
  const int tid = threadIdx.x, wg = blockIdx.x, c = wg / A.split, part = wg - c * A.split;
  double sums[1], maxs[1], acc[RPT];
  sums[0] = maxs[0] = 0.;
#pragma unroll
  for (int j = 0; j < RPT; ++j) acc[j] = 0.;
  unsigned long long tt[T_N];
#pragma unroll
  for (int i = 0; i < T_N; ++i) tt[i] = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  unsigned long long t0 = t_begin, t1;
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
      if (((q >> 6) << 8) < S.pi.y) { S.ic[i] = i4[q]; S.va[i] = v2[2 * q]; S.vb[i] = v2[2 * q + 1]; }
    }
    S.meta = A.meta[((size_t)g * kCsThreads + tid) * (RPT == 16 ? 2 : 1)]; if (RPT == 16) S.meta1 = A.meta[((size_t)g * kCsThreads + tid) * 2 + 1];
  };
  auto gather = [&](const CsSet<NQ> &S) {
    const double *xb = x + S.pi.x;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if ((((tid + i * kCsThreads) >> 6) << 8) < S.pi.y) {
        xg[i][0] = xb[S.ic[i].x >> kCsSlotBits];
        xg[i][1] = xb[S.ic[i].y >> kCsSlotBits];
        xg[i][2] = xb[S.ic[i].z >> kCsSlotBits];
        xg[i][3] = xb[S.ic[i].w >> kCsSlotBits];
      }
    }
  };
#define STAMP(slot) do { t1 = __builtin_amdgcn_s_memtime(); tt[slot] += t1 - t0; t0 = t1; } while (0)
  auto step = [&](int g, CsSet<NQ> &X, CsSet<NQ> &Y, int buf) {
    double *pb = prod[buf];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if ((((tid + i * kCsThreads) >> 6) << 8) < X.pi.y) {
        pb[X.ic[i].x & (kCsPass - 1)] = X.va[i].x * xg[i][0];
        pb[X.ic[i].y & (kCsPass - 1)] = X.va[i].y * xg[i][1];
        pb[X.ic[i].z & (kCsPass - 1)] = X.vb[i].x * xg[i][2];
        pb[X.ic[i].w & (kCsPass - 1)] = X.vb[i].y * xg[i][3];
      }
    }
    const unsigned long long mc = X.meta;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the scatter has left the wave
    STAMP(T_WAIT_SCATTER);
    __syncthreads();
    STAMP(T_BARRIER);
    if (g + 1 < g1) gather(Y);
    if (g + 2 < g1) load(g + 2, X);
    STAMP(T_ISSUE);
    cs_row_sums<RPT>(pb, mc, RPT == 16 ? X.meta1 : 0ull, acc);
    STAMP(T_ROWSUM);
  };
  if (g0 < g1) {
    load(g0, S0);
    gather(S0);
    if (g0 + 1 < g1) load(g0 + 1, S1);
  }
  STAMP(T_PRO);
  for (int g = g0; g < g1; g += 2) {
    step(g, S0, S1, 0);
    if (g + 1 < g1) step(g + 1, S1, S0, 1);
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int rl = j * kCsThreads + tid, r = c * A.R + rl;
    if (rl < A.R && r < A.rows) cs_epilogue(epi, A.split, part, r, acc[j], sums, maxs);
  }
  __builtin_amdgcn_s_waitcnt(0x0000);
  STAMP(T_EPI);
  tt[T_TOTAL] = t1 - t_begin;
  if ((tid & 63) == 0) {
    unsigned long long *o = tl + ((size_t)wg * (kCsThreads / 64) + (tid >> 6)) * T_N;
#pragma unroll
    for (int i = 0; i < T_N; ++i) o[i] = tt[i];
  }
}

struct EpiRaw2 {
  double *y0, *y1;
  static constexpr int kSums = 0, kMaxs = 0;
  __device__ void operator()(int r, double s, double *, double *) const { y0[r] = s; }
  __device__ void split(int r, double s, int part, double *, double *) const { (part ? y1 : y0)[r] = s; }
};

struct DevCs {
  int *passptr; int2 *pinfo; unsigned *idx; double *val; unsigned long long *meta; CsView v; HostCs hc;
  double *scratch = nullptr; unsigned *ticket = nullptr;
  bool build(const Csr &M, int rpt, int split, bool combine = false) {
    if (!build_cs(M.rowptr.data(), M.col.data(), M.val.data(), M.rows, M.cols, hc, rpt, split)) return false;
    passptr = to_dev(hc.passptr); pinfo = to_dev(hc.pinfo); idx = to_dev(hc.idx); val = to_dev(hc.val); meta = to_dev(hc.meta);
    v = CsView{passptr, pinfo, idx, val, meta, hc.rows, hc.cols, hc.nchunks, hc.R, hc.npass, hc.rpt, hc.split};
    if (combine && split > 1) {
      HIP_CHECK(hipMalloc(&scratch, (size_t)hc.nchunks * split * kCsThreads * hc.rpt * 8));
      HIP_CHECK(hipMalloc(&ticket, (size_t)hc.nchunks * 4));
      HIP_CHECK(hipMemset(ticket, 0, (size_t)hc.nchunks * 4));
      v.scratch = scratch; v.ticket = ticket;
    }
    return true;
  }
  void free() { hipFree(passptr); hipFree(pinfo); hipFree(idx); hipFree(val); hipFree(meta); }
};

template <int RPT>
static void timeline(const char *tag, DevCs &D, const double *dx, double *dy, double *dy1) {
  const int nwg = D.hc.nchunks * D.hc.split, nw = kCsThreads / 64;
  unsigned long long *tl;
  HIP_CHECK(hipMalloc(&tl, (size_t)nwg * nw * T_N * 8));
  auto launch = [&] { hipLaunchKernelGGL((k_cs_timed<EpiRaw2, RPT>), dim3(nwg), dim3(kCsThreads), 0, 0, D.v, dx, EpiRaw2{dy, dy1}, tl); };
  const double us = time_us(launch, 10);
  std::vector<unsigned long long> h((size_t)nwg * nw * T_N);
  HIP_CHECK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
  double avg[T_N] = {0}, mx[T_N] = {0};
  for (int w = 0; w < nwg * nw; ++w)
    for (int i = 0; i < T_N; ++i) { avg[i] += (double)h[(size_t)w * T_N + i]; mx[i] = std::max(mx[i], (double)h[(size_t)w * T_N + i]); }
  for (int i = 0; i < T_N; ++i) avg[i] /= (double)nwg * nw;
  // s_memtime ticks at 100 MHz on gfx9 (constant clock): 10 ns per tick
  std::printf("  %-30s instrumented launch %.1f us; per-wave average of the phase sums in us (max over waves):\n", tag, us);
  const char *names[T_N] = {"prologue", "wait gathers+scatter", "barrier", "issue gathers/loads", "row sums", "epilogue", "total"};
  for (int i = 0; i < T_N; ++i) std::printf("      %-22s %8.2f  (%8.2f)\n", names[i], avg[i] * 0.01, mx[i] * 0.01);
  hipFree(tl);
}

static void bench_matrix(const char *name, const Csr &M, int split) {
  std::printf("%s: %d x %d, nnz %d, split %d\n", name, M.rows, M.cols, M.rowptr[M.rows], split);
  std::vector<double> x(M.cols), ref(M.rows);
  std::mt19937_64 g(7);
  std::normal_distribution<double> nd;
  for (auto &v : x) v = nd(g);
  for (int r = 0; r < M.rows; ++r) {
    double s = 0.;
    for (int p = M.rowptr[r]; p < M.rowptr[r + 1]; ++p) s += M.val[p] * x[M.col[p]];
    ref[r] = s;
  }
  double *dx = to_dev(x), *dy, *dy1;
  HIP_CHECK(hipMalloc(&dy, M.rows * sizeof(double)));
  HIP_CHECK(hipMalloc(&dy1, M.rows * sizeof(double)));
  DevCs D;
  if (!D.build(M, 0, split)) { std::printf("  build failed\n"); return; }
  std::printf("  R=%d rpt=%d wgs=%d passes=%d\n", D.hc.R, D.hc.rpt, D.hc.nchunks * D.hc.split, D.hc.npass);
  auto launch = [&] { launch_spmv_cs(D.v, dx, EpiRaw2{dy, dy1}, nullptr, 0, nullptr); };
  std::printf("  shipped k_spmv_cs_ga: %.1f us", time_us(launch, 20));
  if (split == 1) std::printf("  mismatches %ld", mismatches(dy, ref));
  std::printf("\n");
  {
    HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
    auto l2 = [&] {
      const dim3 gg(D.hc.nchunks * D.hc.split), bb(kCsThreads);
      if (D.hc.rpt == 8) hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr);
      else if (D.hc.rpt == 4) hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 4>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr);
      else hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 16>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr);
    };
    std::printf("  braided k_spmv_cs_il: %.1f us", time_us(l2, 20));
    if (split == 1) std::printf("  mismatches %ld", mismatches(dy, ref));
    else {
      std::vector<double> h0(ref.size()), h1(ref.size());
      HIP_CHECK(hipMemcpy(h0.data(), dy, ref.size() * 8, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(h1.data(), dy1, ref.size() * 8, hipMemcpyDeviceToHost));
      double err = 0, scl = 0;
      for (size_t i = 0; i < ref.size(); ++i) { err = std::max(err, std::fabs(h0[i] + h1[i] - ref[i])); scl = std::max(scl, std::fabs(ref[i])); }
      std::printf("  max err %.2e (scale %.1f)", err, scl);
    }
    std::printf("\n");
  }
  {
    const int nwg = D.hc.nchunks * D.hc.split, nw = kCsThreads / 64;
    std::vector<unsigned long long> h((size_t)256 * 16 * 8);
    HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(cs_lab_tl), h.size() * 8));
    double avg[7] = {0}, mx[7] = {0};
    for (int w = 0; w < nwg * nw; ++w)
      for (int i = 0; i < 7; ++i) { avg[i] += (double)h[(size_t)w * 8 + i]; mx[i] = std::max(mx[i], (double)h[(size_t)w * 8 + i]); }
    const char *names[7] = {"prologue", "wait gathers+scatter", "barrier", "braid", "epilogue", "-", "total"};
    std::printf("    braided timeline, kcycles per wave: ");
    for (int i = 0; i < 7; ++i) if (i != 5) std::printf("%s %.1f (max %.1f)  ", names[i], avg[i] / (nwg * nw) * 1e-3, mx[i] * 1e-3);
    std::printf("\n");
  }
  if (D.hc.rpt == 8) {
    const dim3 gg(D.hc.nchunks * D.hc.split), bb(kCsThreads);
    auto a1 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 1>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    auto a2 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 2>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    auto g1 = [&] { hipLaunchKernelGGL((k_spmv_cs_ga<EpiRaw2, 8, 1>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    auto g2 = [&] { hipLaunchKernelGGL((k_spmv_cs_ga<EpiRaw2, 8, 2>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    std::printf("    ablations  il: gathers from 2 KB table %.1f us, no row sums %.1f us;  ga: %.1f / %.1f us\n", time_us(a1, 20), time_us(a2, 20),
                time_us(g1, 20), time_us(g2, 20));
    auto a0 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 0>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    auto a6 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 6>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    auto a7 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 7>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    auto a8 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 8>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    auto a9 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 9>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    auto print_tl = [&](const char *tag) {
      const int nwg = D.hc.nchunks * D.hc.split, nw = kCsThreads / 64;
      HIP_CHECK(hipDeviceSynchronize());
      std::vector<unsigned long long> h((size_t)256 * 16 * 8);
      HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(cs_lab_tl), h.size() * 8));
      double avg[7] = {0}, mx[7] = {0};
      for (int w = 0; w < nwg * nw; ++w)
        for (int i = 0; i < 7; ++i) { avg[i] += (double)h[(size_t)w * 8 + i]; mx[i] = std::max(mx[i], (double)h[(size_t)w * 8 + i]); }
      const char *names[7] = {"prologue", "wait gathers+scatter(+pre-barrier issue)", "barrier", "braid", "epilogue", "-", "total"};
      std::printf("      %s timeline, kcycles per wave: ", tag);
      for (int i = 0; i < 7; ++i) if (i != 5) std::printf("%s %.1f (max %.1f)  ", names[i], avg[i] / (nwg * nw) * 1e-3, mx[i] * 1e-3);
      std::printf("\n");
    };
    auto a10 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 10>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    {
      const double t = time_us(a10, 30), t6 = time_us(a6, 30), tb = time_us(a10, 30), t6b = time_us(a6, 30);
      std::printf("    round 4: ABL 10 (stream before the barrier, one gather per row slot): %.1f us vs ABL 6 %.1f us (again %.1f / %.1f)\n", t, t6, tb, t6b);
    }
    auto a12 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 12>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    {
      HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
      const double t = time_us(a12, 30), t6 = time_us(a6, 30), tb = time_us(a12, 30), t6b = time_us(a6, 30);
      a12();
      HIP_CHECK(hipDeviceSynchronize());
      std::printf("    round 4: ABL 12 (as 6, straight-line tail steps): %.1f us vs ABL 6 %.1f us (again %.1f / %.1f)", t, t6, tb, t6b);
      if (split == 1) std::printf("  mismatches %ld", mismatches(dy, ref));
      std::printf("\n");
    }
    auto a14 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 14>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    {
      HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
      const double t = time_us(a14, 30), t6 = time_us(a6, 30), tb = time_us(a14, 30), t6b = time_us(a6, 30);
      a14();
      HIP_CHECK(hipDeviceSynchronize());
      std::printf("    round 4: ABL 14 (as 6, prologue: index quads, gathers, then values): %.1f us vs ABL 6 %.1f us (again %.1f / %.1f)", t, t6, tb, t6b);
      if (split == 1) std::printf("  mismatches %ld", mismatches(dy, ref));
      std::printf("\n");
    }
    for (int v = 7; v <= 9; ++v) {
      HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
      auto run = [&] { if (v == 7) a7(); else if (v == 8) a8(); else a9(); };
      const double t = time_us(run, 30), t0 = time_us(a0, 30), tb = time_us(run, 30);
      run();
      HIP_CHECK(hipDeviceSynchronize());
      std::printf("    round 4: ABL %d (%s): %.1f us vs %.1f us plain (again %.1f)", v,
                  v == 7 ? "stream + half the gathers before the barrier" : v == 8 ? "stream before the barrier, all gathers in the first row slot" : "stream + all gathers before the barrier",
                  t, t0, tb);
      if (split == 1) std::printf("  mismatches %ld", mismatches(dy, ref));
      std::printf("\n");
      print_tl(v == 7 ? "ABL 7" : v == 8 ? "ABL 8" : "ABL 9");
    }
    a0(); print_tl("plain");
    a6(); print_tl("ABL 6");
    {
      HIP_CHECK(hipMemset(dy, 0xff, ref.size() * 8));
      const double t6 = time_us(a6, 30), t0 = time_us(a0, 30), t6b = time_us(a6, 30), t0b = time_us(a0, 30);
      a6();
      HIP_CHECK(hipDeviceSynchronize());
      std::printf("    round 4: stream loads issued BEFORE the barrier (ABL 6): %.1f us vs %.1f us plain (interleaved A/B: %.1f / %.1f)", t6, t0, t6b, t0b);
      if (split == 1) std::printf("  mismatches %ld", mismatches(dy, ref));
      std::printf("\n");
    }
    auto a5 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 8, 5>), gg, bb, 0, 0, D.v, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
    std::printf("    round 4: pass stream with the non-temporal policy (nt): %.1f us vs %.1f us plain (interleaved A/B: %.1f / %.1f)\n", time_us(a5, 30), time_us(a0, 30),
                time_us(a5, 30), time_us(a0, 30));
  }
  if (getenv("LAB_TIMELINE") == nullptr) { D.free(); hipFree(dx); hipFree(dy); hipFree(dy1); return; }
  if (D.hc.rpt == 8) timeline<8>("k_cs_timed<8>", D, dx, dy, dy1);
  else if (D.hc.rpt == 4) timeline<4>("k_cs_timed<4>", D, dx, dy, dy1);
  else if (D.hc.rpt == 16) timeline<16>("k_cs_timed<16>", D, dx, dy, dy1);
  D.free();
  hipFree(dx); hipFree(dy); hipFree(dy1);
}

// in-kernel combine: correctness of EVERY row over many launches (stale partials of the previous launch sit in the L2s),
// with changing x so that a stale read cannot go unnoticed
static void bench_combine(const char *name, const Csr &M, int split) {
  DevCs D;
  if (!D.build(M, 0, split, true)) { std::printf("%s split %d: build failed\n", name, split); return; }
  std::printf("%s: combine mode split %d  R=%d rpt=%d wgs=%d passes=%d\n", name, split, D.hc.R, D.hc.rpt, D.hc.nchunks * split, D.hc.npass);
  std::mt19937_64 g(11);
  std::normal_distribution<double> nd;
  double *dy; HIP_CHECK(hipMalloc(&dy, M.rows * sizeof(double)));
  long bad_total = 0; double worst = 0;
  std::vector<double> x(M.cols), ref(M.rows), h(M.rows);
  double *dx; HIP_CHECK(hipMalloc(&dx, M.cols * 8));
  for (int rep = 0; rep < 6; ++rep) {
    for (auto &v : x) v = nd(g) * (rep + 1);
    HIP_CHECK(hipMemcpy(dx, x.data(), M.cols * 8, hipMemcpyHostToDevice));
    for (int r = 0; r < M.rows; ++r) { double s = 0.; for (int p = M.rowptr[r]; p < M.rowptr[r + 1]; ++p) s += M.val[p] * x[M.col[p]]; ref[r] = s; }
    HIP_CHECK(hipMemset(dy, 0xff, M.rows * 8));
    for (int k = 0; k < 3; ++k) launch_spmv_cs(D.v, dx, EpiStore{dy, 0}, nullptr, 0, nullptr);
    HIP_CHECK(hipMemcpy(h.data(), dy, M.rows * 8, hipMemcpyDeviceToHost));
    double scl = 0, err = 0;
    for (int r = 0; r < M.rows; ++r) { scl = std::max(scl, std::fabs(ref[r])); const double e = std::fabs(h[r] - ref[r]); if (!(e <= 1e300)) err = 1e300; else err = std::max(err, e); }
    long bad = 0;
    for (int r = 0; r < M.rows; ++r) bad += !(std::fabs(h[r] - ref[r]) <= 1e-12 * scl);
    bad_total += bad; worst = std::max(worst, err / scl);
  }
  auto launch = [&] { launch_spmv_cs(D.v, dx, EpiStore{dy, 0}, nullptr, 0, nullptr); };
  std::printf("  braided + in-kernel combine: %.1f us   rows off by > 1e-12: %ld   worst rel err %.2e\n", time_us(launch, 30), bad_total, worst);
  {
    const int nwg = D.hc.nchunks * D.hc.split, nw = kCsThreads / 64;
    std::vector<unsigned long long> hh((size_t)256 * 16 * 8);
    HIP_CHECK(hipMemcpyFromSymbol(hh.data(), HIP_SYMBOL(cs_lab_tl), hh.size() * 8));
    double avg[7] = {0}, mx[7] = {0};
    for (int w = 0; w < nwg * nw; ++w)
      for (int i = 0; i < 7; ++i) { avg[i] += (double)hh[(size_t)w * 8 + i]; mx[i] = std::max(mx[i], (double)hh[(size_t)w * 8 + i]); }
    const char *names[7] = {"prologue", "wait+scatter", "barrier", "braid", "combine+epilogue", "-", "total"};
    std::printf("    timeline, kcycles per wave: ");
    for (int i = 0; i < 7; ++i) if (i != 5) std::printf("%s %.1f (max %.1f)  ", names[i], avg[i] / (nwg * nw) * 1e-3, mx[i] * 1e-3);
    std::printf("\n");
  }
  {  // the same layout without the combine: partial outputs
    CsView v2 = D.v; v2.scratch = nullptr; v2.ticket = nullptr;
    double *dy1; HIP_CHECK(hipMalloc(&dy1, M.rows * 8));
    auto l2 = [&] { launch_spmv_cs(v2, dx, EpiRaw2{dy, dy1}, nullptr, 0, nullptr); };
    if (split == 2) std::printf("    same layout, partial outputs (no combine): %.1f us\n", time_us(l2, 30));
    const dim3 gg(D.hc.nchunks * D.hc.split), bb(kCsThreads);
    if (D.hc.rpt == 16) {
      auto a0 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 16, 0>), gg, bb, 0, 0, v2, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
      auto a1 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 16, 1>), gg, bb, 0, 0, v2, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
      auto a2 = [&] { hipLaunchKernelGGL((k_spmv_cs_il<EpiRaw2, 16, 2>), gg, bb, 0, 0, v2, dx, EpiRaw2{dy, dy1}, nullptr, nullptr); };
      std::printf("    no combine (parts write to 2 vectors): %.1f us; gathers from 2 KB table %.1f us; no row sums %.1f us\n", time_us(a0, 20), time_us(a1, 20), time_us(a2, 20));
    }
    hipFree(dy1);
  }
  D.free(); hipFree(dx); hipFree(dy);
}

int main(int argc, char **argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 2000000, n = argc > 2 ? atoi(argv[2]) : 1000000;
  const int k = argc > 3 ? atoi(argv[3]) : 20;
  Csr At;
  At.rows = n; At.cols = m;
  At.rowptr.resize(n + 1);
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
  if (getenv("LAB_BASE")) {
    bench_matrix("K1 shape  CSR(A)", Ar, 1);
    bench_matrix("K2 shape  CSR(A')", At, 2);
  }
  if (getenv("LAB_SKIP_COMBINE")) return 0;
  bench_combine("K1 shape  CSR(A)", Ar, 2);
  bench_combine("K2 shape  CSR(A')", At, 2);
  bench_combine("K2 shape  CSR(A')", At, 4);
  if (Ar.rows <= 1000000) bench_combine("K1 shape  CSR(A)", Ar, 4);
  return 0;
}
