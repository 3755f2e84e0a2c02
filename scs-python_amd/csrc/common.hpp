// common.hpp — error handling, device buffers and deterministic block reductions
// shared by every kernel of libscs_hip (gfx950 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/scs_hip.h"
#include "options.hpp"

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

// Process-wide cache of device blocks.  hipFree synchronises the device and unmaps (0.65 ms per scs_finish of a config-5 workspace,
// 0.3-0.5 s per batch of 512), hipMalloc maps: a workspace that dies hands its blocks here instead — only from ~ScsHipWork, which has
// synchronised its stream first (t_pool_release; temporaries that die while their stream is still busy keep using hipFree, whose
// implicit synchronisation is what makes that safe) — and every allocation looks here first (exact size, same device).
// Bounded: SCS_HIP_POOL_MB, default 1024 (round 5, ADVICE r04: the 16 GiB of round 4 starved torch / RCCL and other processes of the
// same GPU, whose failing allocations never reach trim() here; bench.py asks for 16384 itself: its batch legs tear down 512 workspaces
// at a time); 0 disables; emptied when a hipMalloc of this library fails and by scs_hip_trim_pool().  Recycled blocks carry the
// previous owner's data where driver-fresh pages were zero: every DevBuf user either writes its buffer before reading it or asks
// for alloc_zero (which clears recycled blocks); SCS_HIP_POOL_POISON=1 fills every recycled block with NaN bit patterns first
// (synchronously — a debug mode for running the test suite against that assumption).
struct DevPool {
  std::mutex m;
  std::vector<std::pair<std::pair<int, size_t>, void *>> blocks;  // ((device, bytes), pointer)
  size_t cached = 0, cap = 0;
  bool cap_read = false;
  static DevPool &inst() { static DevPool p; return p; }
  size_t capacity() {
    if (!cap_read) { const char *e = getenv("SCS_HIP_POOL_MB"); cap = (size_t)(e ? atol(e) : 1024) << 20; cap_read = true; }  // (process-wide: read once)
    return cap;
  }
  void *get(size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(m);
    for (size_t i = blocks.size(); i-- > 0;)
      if (blocks[i].first.first == dev && blocks[i].first.second == bytes) {
        void *p = blocks[i].second;
        blocks.erase(blocks.begin() + (long)i);
        cached -= bytes;
        if (opts().pool_poison)  // (labs: recycled blocks arrive poisoned) { (void)hipMemset(p, 0xFF, bytes); (void)hipDeviceSynchronize(); }
        return p;
      }
    return nullptr;
  }
  bool put(void *p, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(m);
    if (bytes > capacity()) return false;
    // over the cap: the OLDEST blocks go back to the driver first (a long-lived service with changing problem sizes does not
    // end up holding a cap's worth of blocks nobody asks for any more)
    size_t drop = 0;
    while (drop < blocks.size() && cached + bytes > capacity()) {
      (void)hipFree(blocks[drop].second);
      cached -= blocks[drop].first.second;
      ++drop;
    }
    if (drop) blocks.erase(blocks.begin(), blocks.begin() + (long)drop);
    blocks.push_back({{dev, bytes}, p});
    cached += bytes;
    return true;
  }
  size_t held_bytes() {
    std::lock_guard<std::mutex> lk(m);
    return cached;
  }
  void trim() {
    std::lock_guard<std::mutex> lk(m);
    for (auto &b : blocks) (void)hipFree(b.second);  // (blocks of other devices: hipFree takes any device's pointer)
    blocks.clear();
    cached = 0;
  }
};
static thread_local int t_pool_release = 0;  // > 0: DevBuf / Arena releases on this thread go to the pool (see DevPool)
inline void *dev_malloc(size_t bytes) {
  if (void *p = DevPool::inst().get(bytes)) return p;
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    DevPool::inst().trim();
    e = hipMalloc(&p, bytes);
  }
  HIP_CHECK(e);
  return p;
}
inline void dev_free(void *p, size_t bytes) {
  if (!p) return;
  if (t_pool_release > 0 && DevPool::inst().put(p, bytes)) return;
  (void)hipFree(p);
}

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
  //  first chunk sized from an estimate of the workspace (set by scs_init before the first allocation): a service that
  // keeps thousands of tiny workspaces alive pays kilobytes each, not 4 MiB; later chunks are kChunkBytes
  size_t first_chunk = kChunkBytes;
  Arena() = default;
  Arena(const Arena &) = delete;
  Arena &operator=(const Arena &) = delete;
  ~Arena() {
    for (Chunk &c : chunks) dev_free(c.p, c.size);
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
    if (p && !in_arena) dev_free(p, sizeof(T) * (n ? n : 1));
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
    p = (T *)dev_malloc(bytes);
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
  const size_t want = chunks.empty() ? first_chunk : kChunkBytes;
  Chunk c{nullptr, bytes > want ? bytes : want, 0};
  c.p = (char *)dev_malloc(c.size);
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
