// micro-benchmark: does a cache-policy modifier on the gather loads change the cost of an L2-hit random gather?
// 2e7 random 8-byte gathers from a 1 MB table (64 distinct 128-byte lines per wave instruction), 8 loads in
// flight per lane, load flavours issued through inline asm (one s_waitcnt for the batch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o devtools/gather_flavours tools/gather_flavours.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define LOADS8(MOD)                                                                                                     \
  asm volatile("global_load_dwordx2 %0, %8, off " MOD "\n global_load_dwordx2 %1, %9, off " MOD                         \
               "\n global_load_dwordx2 %2, %10, off " MOD "\n global_load_dwordx2 %3, %11, off " MOD                    \
               "\n global_load_dwordx2 %4, %12, off " MOD "\n global_load_dwordx2 %5, %13, off " MOD                    \
               "\n global_load_dwordx2 %6, %14, off " MOD "\n global_load_dwordx2 %7, %15, off " MOD                    \
               "\n s_waitcnt vmcnt(0)"                                                                                  \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])                 \
               : "memory")

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ idx, const double *__restrict__ tab, double *out, long n) {
  const long base = ((long)blockIdx.x * 256 * 16) + threadIdx.x;
  double acc = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const double *p[8];
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long q = base + (h * 8 + i) * 256;
      p[i] = tab + (q < n ? idx[q] : 0);
    }
    if (MODE == 0) LOADS8("");
    else if (MODE == 1) LOADS8("nt");
    else if (MODE == 2) LOADS8("sc0");
    else if (MODE == 3) LOADS8("sc1");
    else if (MODE == 4) LOADS8("sc0 sc1");
    else LOADS8("sc0 sc1 nt");
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
  }
  if (acc == 123.456) out[0] = acc;
}

template <int MODE>
void run(const char *name, const int *idx, const double *tab, double *out, long n) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = (int)((n + 256L * 16 - 1) / (256L * 16));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), 0, 0, idx, tab, out, n);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), 0, 0, idx, tab, out, n);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
  printf("  %-14s %.1f us\n", name, ms * 1e3);
}

int main() {
  const long n = 20000000;
  for (long tsize : {131072L, 1048576L}) {
    for (int L : {64, 24}) {
      std::vector<int> h(n); std::mt19937 g(2);
      for (long i = 0; i < n; i += 64) {
        int lines[64];
        for (int k = 0; k < L; ++k) lines[k] = (int)(g() % (tsize / 16));
        for (int k = 0; k < 64 && i + k < n; ++k) h[i + k] = lines[k % L] * 16 + (int)(g() % 16);
      }
      int *idx; double *tab; double *out;
      CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&tab, tsize * 8)); CK(hipMalloc(&out, 64));
      CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(tab, 0, tsize * 8));
      printf("%ld MB table, %d distinct lines per wave instruction\n", tsize * 8 >> 20, L);
      run<0>("plain", idx, tab, out, n);
      run<1>("nt", idx, tab, out, n);
      run<2>("sc0", idx, tab, out, n);
      run<3>("sc1", idx, tab, out, n);
      run<4>("sc0 sc1", idx, tab, out, n);
      run<5>("sc0 sc1 nt", idx, tab, out, n);
      CK(hipFree(idx)); CK(hipFree(tab)); CK(hipFree(out));
    }
  }
  return 0;
}
