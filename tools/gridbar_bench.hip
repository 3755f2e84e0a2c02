// micro-benchmark: cost of a software grid barrier (monotonic atomic counter, agent scope) among G co-resident workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);  // agent scope by default for global atomics in HIP
    while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_bar(unsigned *ctr, double *buf, int nbar, int work) {
  double acc = 0;
  for (int b = 1; b <= nbar; ++b) {
    // a little dependent global traffic between barriers: each block writes a slot, reads its neighbour's slot of the previous round
    if (work) {
      buf[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + b;
    }
    grid_barrier(ctr, (unsigned)b * gridDim.x);
    if (work) acc += buf[(size_t)((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
  }
  if (acc == 12345.678) buf[0] = acc;
}

int main() {
  unsigned *ctr; double *buf;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, 1024 * 256 * 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int nbar = 1000;
  for (int work = 0; work < 2; ++work)
    for (int G : {1, 2, 8, 16, 32, 64, 128, 256, 512}) {
      CK(hipMemset(ctr, 0, 4));
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_bar, dim3(G), dim3(256), 0, 0, ctr, buf, nbar, work);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("work=%d G=%4d: %.3f us per barrier\n", work, G, ms * 1e3 / nbar);
    }
  return 0;
}
