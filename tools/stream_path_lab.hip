// stream_path_lab.hip — VERDICT r03 item 1(a): can the 13 B/nnz pass STREAM of K1 / K2 be pulled into L2 by something other than the
// vector L1 of the CU that consumes it?  The only other path from a CU to L2 is the scalar data cache.  This lab measures how many
// 128-byte lines per second ONE helper wavefront per CU can touch (= have L2 fetch from HBM) with
//   (s) s_load_dword at 128-byte strides, never waited for (the hardware's lgkmcnt saturates: at most 15 in flight per wavefront),
//   (v) global_load_dword, one lane per line (64 lines per instruction; through the vector L1 — for reference only: these occupy
//       the very miss queue the experiment wants to relieve),
// against what one K1 launch needs: 106 KB of stream per CU and pass, ~10 passes per 94 us launch = 11.3 GB/s per CU.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_path_lab tools/stream_path_lab.hip && ./gpurun_out/stream_path_lab
#include <cstdio>
#include <hip/hip_runtime.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef const int __attribute__((address_space(4))) *ScalarPtr;

__global__ __launch_bounds__(64) void k_scalar_touch(const int *base, size_t bytes_per_wave, int *sink) {
  const char *p = (const char *)base + (size_t)blockIdx.x * bytes_per_wave;
  int acc = 0;
  // one s_load_dword per 128-byte line; the results are summed only at the end, so the compiler may keep as many loads in flight
  // as the counter allows (it groups them behind one s_waitcnt lgkmcnt(0) per unrolled batch)
  for (size_t o = 0; o < bytes_per_wave; o += 128 * 16) {
    int t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = *(ScalarPtr)(p + o + 128 * k);
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += t[k];
  }
  if (acc == 0x7fffffff) *sink = acc;
}
__global__ __launch_bounds__(64) void k_vector_touch(const int *base, size_t bytes_per_wave, int *sink) {
  const char *p = (const char *)base + (size_t)blockIdx.x * bytes_per_wave + 128 * threadIdx.x;
  int acc = 0;
  for (size_t o = 0; o < bytes_per_wave; o += 128 * 64 * 8) {
    int t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = *(const int *)(p + o + 128 * 64 * k);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += t[k];
  }
  if (acc == 0x7fffffff) *sink = acc;
}

int main() {
  const int waves = 256;                       // one helper wavefront per CU
  const size_t per_wave = 4u << 20;            // 4 MiB each: 1 GiB in all, far beyond the L2s and the Infinity Cache
  int *buf, *sink;
  CK(hipMalloc(&buf, per_wave * waves));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, per_wave * waves));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a, 0));
      if (which == 0) hipLaunchKernelGGL(k_scalar_touch, dim3(waves), dim3(64), 0, 0, buf, per_wave, sink);
      else hipLaunchKernelGGL(k_vector_touch, dim3(waves), dim3(64), 0, 0, buf, per_wave, sink);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double gbs = (double)per_wave * waves / (ms * 1e-3) / 1e9;
      std::printf("%s path, one wavefront per CU: %.3f ms for 1 GiB of lines touched = %.1f GB/s chip-wide = %.2f GB/s per CU (K1 needs 11.3 GB/s per CU)\n",
                  which == 0 ? "scalar (s_load_dword per 128 B)" : "vector (one lane per 128-B line)", ms, gbs, gbs / waves);
    }
  }
  return 0;
}
