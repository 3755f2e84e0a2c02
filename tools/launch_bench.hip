// micro-benchmark: cost of back-to-back dependent kernel launches on one stream (empty kernel / one-load chain / small reduce)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_empty() {}
__global__ void k_flag(const int *fl, double *x) { if (fl[0]) return; x[threadIdx.x + blockIdx.x * blockDim.x] += 1.0; }
__global__ void k_chain(const int *fl, const double *part, double *x, int n) {
  if (fl[0]) return;
  __shared__ double s;
  double a = 0;
  for (int i = threadIdx.x; i < 64; i += blockDim.x) a += part[i];
  if (threadIdx.x == 0) s = a;
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = x[i] * 0.999 + s;
}
int main() {
  int *fl; double *x, *part;
  CK(hipMalloc(&fl, 64)); CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&part, 4096));
  CK(hipMemset(fl, 0, 64)); CK(hipMemset(x, 0, 1 << 20)); CK(hipMemset(part, 0, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 2000;
  for (int variant = 0; variant < 4; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
        else if (variant == 1) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s);
        else if (variant == 2) hipLaunchKernelGGL(k_flag, dim3(6), dim3(256), 0, s, fl, x);
        else hipLaunchKernelGGL(k_chain, dim3(6), dim3(256), 0, s, fl, part, x, 5400);
      }
      auto t1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s));
      auto t2 = std::chrono::steady_clock::now();
      if (rep == 1)
        printf("variant %d: host enqueue %.2f us/launch, end-to-end %.2f us/launch\n", variant,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
  }
  return 0;
}
