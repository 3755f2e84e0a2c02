// xcd_coherence_lab.hip — which load / store flavours hand data from one workgroup to another INSIDE a kernel, and what they cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/xcd_lab tools/xcd_coherence_lab.hip && /tmp/xcd_lab
// Pairs of workgroups (writer, reader) ping-pong a 32 KB buffer `iters` times through two monotonic flags (agent-scope
// relaxed atomics).  The reader checks every value (stale reads are counted) and the round trip is timed.
//   pairing 0: writer id w, reader w + 8  (same XCD if workgroup ids are dealt round-robin to the 8 XCDs)
//   pairing 1: writer 2j, reader 2j + 1   (neighbouring XCDs)
//   store flavour: 0 plain, 1 sc1 (write-through)            load flavour: 0 sc1, 1 plain, 2 plain after buffer_inv sc1, 3 sc0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kN = 4096;  // doubles per buffer
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int ST, int LD>
__global__ __launch_bounds__(256) void k_pingpong(double *bufs, unsigned *flags, int iters, int pairing, long *stale, unsigned *xcc, long long *ticks) {
  const int id = blockIdx.x, npair = gridDim.x / 2;
  int pair, role;
  if (pairing == 0) { pair = id % npair; role = id / npair; } else { pair = id / 2; role = id % 2; }
  double *buf = bufs + (size_t)pair * kN;
  unsigned *f_w = flags + 2 * pair, *f_r = f_w + 1;
  const int tid = threadIdx.x;
  if (tid == 0) xcc[id] = xcc_id();
  long bad = 0;
  const long long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    if (role == 0) {
      for (int i = tid; i < kN; i += 256) {
        const double v = (double)it * 8192. + i;
        if (ST == 0) buf[i] = v; else __hip_atomic_store(&buf[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(f_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(f_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
    } else {
      if (tid == 0)
        while (__hip_atomic_load(f_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      if (LD == 2) asm volatile("buffer_inv sc1" ::: "memory");
      for (int i = tid; i < kN; i += 256) {
        double v;
        if (LD == 0) v = __hip_atomic_load(&buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (LD == 3) { const double *p = &buf[i]; asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
        else v = *(volatile double *)&buf[i];
        if (v != (double)it * 8192. + i) ++bad;
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(f_r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const long long t1 = wall_clock64();
  if (bad) atomicAdd((unsigned long long *)stale, (unsigned long long)bad);
  if (tid == 0 && id == 0) ticks[0] = t1 - t0;
}

template <int ST, int LD>
void run(const char *name, int pairing, int npair) {
  double *bufs; unsigned *flags, *xcc; long *stale; long long *ticks;
  CK(hipMalloc(&bufs, (size_t)npair * kN * 8)); CK(hipMalloc(&flags, npair * 8)); CK(hipMalloc(&xcc, npair * 2 * 4));
  CK(hipMalloc(&stale, 8)); CK(hipMalloc(&ticks, 8));
  CK(hipMemset(bufs, 0, (size_t)npair * kN * 8)); CK(hipMemset(flags, 0, npair * 8)); CK(hipMemset(stale, 0, 8));
  const int iters = 2000;
  hipLaunchKernelGGL((k_pingpong<ST, LD>), dim3(2 * npair), dim3(256), 0, 0, bufs, flags, iters, pairing, stale, xcc, ticks);
  CK(hipDeviceSynchronize());
  long hs; long long ht; std::vector<unsigned> hx(2 * npair);
  CK(hipMemcpy(&hs, stale, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hx.data(), xcc, 2 * npair * 4, hipMemcpyDeviceToHost));
  int same = 0;
  for (int p = 0; p < npair; ++p) same += pairing == 0 ? hx[p] == hx[p + npair] : hx[2 * p] == hx[2 * p + 1];
  printf("%-40s pairing %d: %d/%d pairs on one XCD, stale values %ld of %ld, %.2f us per round trip\n", name, pairing, same, npair, hs,
         (long)iters * kN * npair, ht / 100.0 / iters);
  CK(hipFree(bufs)); CK(hipFree(flags)); CK(hipFree(xcc)); CK(hipFree(stale)); CK(hipFree(ticks));
}

int main() {
  for (int pairing = 0; pairing < 2; ++pairing) {
    const int npair = 8;
    run<1, 0>("store sc1, load sc1", pairing, npair);
    run<0, 0>("store plain, load sc1", pairing, npair);
    run<0, 1>("store plain, load plain", pairing, npair);
    run<0, 2>("store plain, buffer_inv sc1 + plain load", pairing, npair);
    run<0, 3>("store plain, load sc0", pairing, npair);
    run<1, 2>("store sc1, buffer_inv sc1 + plain load", pairing, npair);
  }
  { // ids -> XCD map of a 32-workgroup launch
    printf("with 64 pairs (128 workgroups):\n");
    run<0, 2>("store plain, buffer_inv sc1 + plain load", 0, 64);
    run<0, 3>("store plain, load sc0", 0, 64);
  }
  return 0;
}
