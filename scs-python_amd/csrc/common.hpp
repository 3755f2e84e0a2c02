// common.hpp — error handling, device buffers and deterministic block reductions
// shared by every kernel of libscs_hip (gfx950 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/scs_hip.h"

namespace scship {

constexpr int kWave = 64;

inline void set_last_error(const std::string &s);

#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      std::string _m = std::string("HIP error ") + hipGetErrorString(_e) + " at " + __FILE__ + \
                       ":" + std::to_string(__LINE__) + " (" #expr ")";                        \
      throw std::runtime_error(_m);                                                           \
    }                                                                                         \
  } while (0)

// Owning device buffer (HBM).  Everything the ADMM loop touches lives in these.
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    n = count;
    HIP_CHECK(hipMalloc((void **)&p, sizeof(T) * (count ? count : 1)));
  }
  void alloc_zero(size_t count, hipStream_t s) {
    alloc(count);
    HIP_CHECK(hipMemsetAsync(p, 0, sizeof(T) * (count ? count : 1), s));
  }
  void upload(const T *h, size_t count, hipStream_t s) {
    if (count != n || !p) alloc(count);
    if (count) HIP_CHECK(hipMemcpyAsync(p, h, sizeof(T) * count, hipMemcpyHostToDevice, s));
  }
  void download(T *h, size_t count, hipStream_t s) const {
    if (count) HIP_CHECK(hipMemcpyAsync(h, p, sizeof(T) * count, hipMemcpyDeviceToHost, s));
  }
};

// ---------------------------------------------------------------------------
// Deterministic reductions: fixed shuffle tree inside a wave, fixed order over
// waves through LDS.  No float atomics anywhere (run-to-run bit determinism is
// pinned by R:test/test_scs_coverage.py:2283-2301).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, kWave));
  return v;
}

struct BlockSync {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};

// Reductions over a group of NT consecutive lanes (tid = index inside the group); result valid in tid 0;
// `sm` needs NT/64 doubles private to the group.  sync() must synchronise (at least) the group.
template <int NT, class Sync>
__device__ __forceinline__ double group_sum(double v, double *sm, int tid, Sync sync) {
  const int lane = tid & 63, wid = tid >> 6;
  v = wave_sum(v);
  sync();
  if (lane == 0) sm[wid] = v;
  sync();
  double r = 0.;
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += sm[i];
  }
  return r;
}
template <int NT, class Sync>
__device__ __forceinline__ double group_max(double v, double *sm, int tid, Sync sync) {
  const int lane = tid & 63, wid = tid >> 6;
  v = wave_max(v);
  sync();
  if (lane == 0) sm[wid] = v;
  sync();
  double r = 0.;
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r = fmax(r, sm[i]);
  }
  return r;
}
// the whole workgroup as one group
template <int NT>
__device__ __forceinline__ double block_sum(double v, double *sm) {
  return group_sum<NT>(v, sm, (int)threadIdx.x, BlockSync{});
}
template <int NT>
__device__ __forceinline__ double block_max(double v, double *sm) {
  return group_max<NT>(v, sm, (int)threadIdx.x, BlockSync{});
}

// run-ahead mode of the ADMM loop (vec.hpp, F_STALL): kernels queued behind a stalled CG solve return at once
#define SCS_STALL_GUARD(stall) do { if ((stall) && *(stall)) return; } while (0)

// abs that propagates NaN into max-reductions as +inf (so a NaN iterate never passes a tolerance test)
__device__ __forceinline__ double abs_nan_inf(double x) { return (x != x) ? INFINITY : fabs(x); }

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace scship
