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

// Workspace arena: small problems (a batch of them: BASELINE.json configs[4]) pay more for the ~100 hipMalloc /
// hipMemset / hipFree calls of a workspace than for the kernels of scs_init (measured: 3.9 ms per scs_init and 2.5 ms per
// scs_finish of a config-5 problem, 512 of them per batch).  While an arena is current on the calling thread, DevBuf
// takes its memory from the arena's chunks: one hipMalloc + one memset per chunk, nothing to free per buffer (the
// arena returns its chunks when the workspace dies), and a zero-initialised buffer costs nothing because fresh arena
// memory is zero and is handed out once.  Large allocations bypass it (a big problem keeps exact-size buffers).
struct Arena {
  struct Chunk { char *p; size_t size, used; };
  std::vector<Chunk> chunks;
  hipStream_t stream = nullptr;  // chunks are zeroed on this stream: buffers must first be used on it (they are)
  static constexpr size_t kChunkBytes = 4u << 20, kMaxAlloc = 1u << 20, kAlign = 256;
  Arena() = default;
  Arena(const Arena &) = delete;
  Arena &operator=(const Arena &) = delete;
  ~Arena() {
    for (Chunk &c : chunks) (void)hipFree(c.p);
  }
  void *take(size_t bytes);
};
static thread_local Arena *t_arena = nullptr;
struct ArenaScope {  // makes `a` the calling thread's arena for the lifetime of the scope (nullptr: none)
  Arena *prev;
  explicit ArenaScope(Arena *a) : prev(t_arena) { t_arena = a; }
  ~ArenaScope() { t_arena = prev; }
};

// Owning device buffer (HBM).  Everything the ADMM loop touches lives in these.
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  bool in_arena = false;  // memory belongs to the workspace arena: nothing to free here
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && !in_arena) (void)hipFree(p);
    p = nullptr;
    n = 0;
    in_arena = false;
  }
  void alloc(size_t count) {
    release();
    n = count;
    const size_t bytes = sizeof(T) * (count ? count : 1);
    if (t_arena && bytes <= Arena::kMaxAlloc) {
      p = (T *)t_arena->take(bytes);
      in_arena = true;
      return;
    }
    HIP_CHECK(hipMalloc((void **)&p, bytes));
  }
  void alloc_zero(size_t count, hipStream_t s) {
    alloc(count);
    if (in_arena && t_arena && s == t_arena->stream) return;  // fresh arena memory is zero (zeroed on this stream)
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

inline void *Arena::take(size_t bytes) {
  bytes = (bytes + kAlign - 1) / kAlign * kAlign;
  for (Chunk &c : chunks)
    if (c.size - c.used >= bytes) {
      void *r = c.p + c.used;
      c.used += bytes;
      return r;
    }
  Chunk c{nullptr, bytes > kChunkBytes ? bytes : kChunkBytes, 0};
  HIP_CHECK(hipMalloc((void **)&c.p, c.size));
  HIP_CHECK(hipMemsetAsync(c.p, 0, c.size, stream));
  c.used = bytes;
  chunks.push_back(c);
  return c.p;
}

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
