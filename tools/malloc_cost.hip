// cost of the device allocations of a config-5 batch with the dense linear solver (lab): 512 x 14.6 MB from T host threads against ONE slab
// hipcc --offload-arch=gfx950 -O2 -o malloc_cost malloc_cost.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  (void)hipFree(0);
  const size_t bytes = (size_t)1360 * 1360 * 8;  // G^-1 of one member (NP = 1360)
  const int N = 512;
  for (int T : {1, 4, 16}) {
    std::vector<void *> p(N, nullptr);
    double t = now();
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k)
      th.emplace_back([&, k] {
        (void)hipSetDevice(0);
        for (int i = k; i < N; i += T) (void)hipMalloc(&p[i], bytes);
      });
    for (auto &x : th) x.join();
    const double a = now() - t;
    t = now();
    for (int i = 0; i < N; ++i) (void)hipMemsetAsync(p[i], 0, bytes, 0);
    (void)hipDeviceSynchronize();
    const double m = now() - t;
    t = now();
    for (int i = 0; i < N; ++i) (void)hipFree(p[i]);
    printf("%2d threads: %d x hipMalloc(%.1f MB) %.1f ms (%.3f ms each), first touch (memset) %.1f ms, hipFree %.1f ms\n", T, N, bytes / 1e6, a, a / N, m, now() - t);
  }
  for (int rep = 0; rep < 2; ++rep) {
    void *slab = nullptr;
    double t = now();
    (void)hipMalloc(&slab, bytes * N);
    const double a = now() - t;
    t = now();
    (void)hipMemsetAsync(slab, 0, bytes * N, 0);
    (void)hipDeviceSynchronize();
    const double m = now() - t;
    t = now();
    (void)hipFree(slab);
    printf("one slab of %.2f GB: hipMalloc %.1f ms, first touch %.1f ms, hipFree %.1f ms\n", bytes * N / 1e9, a, m, now() - t);
  }
  return 0;
}
