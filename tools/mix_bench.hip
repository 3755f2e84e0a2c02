// micro-benchmark: do L2-hit gathers and HBM streaming overlap when they run on different workgroups / in the same lanes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: gather only (idx streamed + x gathered); 1: stream only; 2: even blocks gather, odd blocks stream;
// 3: every block does both (interleaved in the same lanes)
template <int MODE>
__global__ __launch_bounds__(256) void k_mix(const int *__restrict__ idx, const double *__restrict__ tab, const double2 *__restrict__ str,
                                             double *out, long n_g, long n_s2) {
  constexpr int PER = 8;
  double acc = 0;
  const int nb = MODE == 2 ? gridDim.x / 2 : gridDim.x;
  const int b = MODE == 2 ? ((blockIdx.x >> 4) << 3) + (blockIdx.x & 7) : blockIdx.x;
  const bool do_g = MODE == 0 || MODE == 3 || (MODE == 2 && ((blockIdx.x >> 3) & 1) == 0);
  const bool do_s = MODE == 1 || MODE == 3 || (MODE == 2 && ((blockIdx.x >> 3) & 1) == 1);
  if (do_g) {
    for (long base = (long)b * 256 * PER + threadIdx.x; base < n_g; base += (long)nb * 256 * PER) {
      int id[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) id[i] = (base + i * 256 < n_g) ? idx[base + i * 256] : 0;
#pragma unroll
      for (int i = 0; i < PER; ++i) acc += tab[id[i]];
    }
  }
  if (do_s) {
    for (long base = (long)b * 256 * PER + threadIdx.x; base < n_s2; base += (long)nb * 256 * PER) {
      double2 v[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) v[i] = (base + i * 256 < n_s2) ? str[base + i * 256] : double2{0, 0};
#pragma unroll
      for (int i = 0; i < PER; ++i) acc += v[i].x + v[i].y;
    }
  }
  if (acc == 123.456) out[0] = acc;
}

template <int MODE>
void run(const char *name, int blocks, const int *idx, const double *tab, const double2 *str, double *out, long n_g, long n_s2) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_mix<MODE>), dim3(blocks), dim3(256), 0, 0, idx, tab, str, out, n_g, n_s2);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_mix<MODE>), dim3(blocks), dim3(256), 0, 0, idx + (long)(i % 8) * n_g, tab, str + (long)(i % 8) * n_s2, out, n_g, n_s2);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
  printf("  %-44s blocks=%5d  %.1f us\n", name, blocks, ms * 1e3);
}

int main() {
  const long n = 20000000, tsize = 131072, n_s2 = 160000000 / 16 * 1;  // 160 MB of double2
  std::vector<int> h(n); std::mt19937 g(1);
  for (long i = 0; i < n; ++i) h[i] = (int)(g() % tsize);
  int *idx; double *tab, *out; double2 *str;
  CK(hipMalloc(&idx, 8 * n * 4)); CK(hipMalloc(&tab, tsize * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&str, 8 * n_s2 * 16));
  for (int k = 0; k < 8; ++k) CK(hipMemcpy(idx + k * n, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(tab, 0, tsize * 8)); CK(hipMemset(str, 0, 8 * n_s2 * 16));
  for (int blocks : {1024, 2048, 4096}) {
    run<0>("gather 2e7 from 1 MB (+80 MB idx)", blocks, idx, tab, str, out, n, n_s2);
    run<1>("stream 160 MB", blocks, idx, tab, str, out, n, n_s2);
    run<2>("both, different workgroups", blocks * 2, idx, tab, str, out, n, n_s2);
    run<3>("both, same lanes", blocks, idx, tab, str, out, n, n_s2);
  }
  return 0;
}
