// micro-benchmark: random 8-byte gather rate from an L2-resident table, several load flavours
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int PER>
__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ idx, const double* __restrict__ tab, double* out, long n) {
  long base = ((long)blockIdx.x * 256 * PER) + threadIdx.x;
  double acc = 0;
  int id[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) id[i] = (base + i * 256 < n) ? idx[base + i * 256] : 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    double v;
    if (MODE == 0) v = tab[id[i]];
    else if (MODE == 1) v = __builtin_nontemporal_load(&tab[id[i]]);
    else if (MODE == 2) { const double* p = &tab[id[i]]; asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
    else if (MODE == 3) { const double* p = &tab[id[i]]; asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
    else { v = (double)__builtin_nontemporal_load((const float*)&tab[id[i]]); }
    acc += v;
  }
  if (acc == 123.456) out[0] = acc;
}

template <int MODE, int PER>
void run(const char* name, const int* idx, const double* tab, double* out, long n) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  int blocks = (int)((n + 256L * PER - 1) / (256L * PER));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_gather<MODE, PER>), dim3(blocks), dim3(256), 0, 0, idx, tab, out, n);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_gather<MODE, PER>), dim3(blocks), dim3(256), 0, 0, idx, tab, out, n);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
  printf("  %-28s per=%d  %.1f us   %.1f Ggather/s\n", name, PER, ms * 1e3, n / (ms * 1e-3) / 1e9);
}

int main() {
  const long n = 20000000;
  // (a) table-size sweep, uniformly random indices
  for (long tsize : {1024L, 2048L, 4096L, 16384L, 131072L}) {
    std::vector<int> h(n); std::mt19937 g(1);
    for (long i = 0; i < n; ++i) h[i] = (int)(g() % tsize);
    int* idx; double* tab; double* out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&tab, tsize * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(tab, 0, tsize * 8));
    printf("table %ld entries (%.3f MB), %ld gathers\n", tsize, tsize * 8 / 1e6, n);
    run<0, 8>("plain", idx, tab, out, n);
    run<0, 16>("plain", idx, tab, out, n);
    CK(hipFree(idx)); CK(hipFree(tab)); CK(hipFree(out));
  }
  // (b) 1 MB table, every group of 64 consecutive gathers (one wave instruction) touches only L distinct 128-byte lines
  for (int L : {64, 32, 16, 8, 4}) {
    const long tsize = 131072;
    std::vector<int> h(n); std::mt19937 g(2);
    for (long i = 0; i < n; i += 64) {
      int lines[64];
      for (int k = 0; k < L; ++k) lines[k] = (int)(g() % (tsize / 16));
      for (int k = 0; k < 64 && i + k < n; ++k) h[i + k] = lines[k % L] * 16 + (int)(g() % 16);
    }
    int* idx; double* tab; double* out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&tab, tsize * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(tab, 0, tsize * 8));
    printf("1 MB table, %d distinct lines per wave instruction\n", L);
    run<0, 8>("plain", idx, tab, out, n);
    run<0, 16>("plain", idx, tab, out, n);
    CK(hipFree(idx)); CK(hipFree(tab)); CK(hipFree(out));
  }
  return 0;
}
