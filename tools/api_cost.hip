// cost of the HIP runtime calls a workspace makes at scs_init / scs_finish (lab): hipcc --offload-arch=gfx950 -O2 -o api_cost api_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int N = 200;
  (void)hipFree(0);
  std::vector<void *> p(N);
  std::vector<hipStream_t> st(N);
  std::vector<hipEvent_t> ev(N);
  double t = now();
  for (int i = 0; i < N; ++i) (void)hipHostMalloc(&p[i], 4096);
  double a = now() - t; t = now();
  for (int i = 0; i < N; ++i) (void)hipHostFree(p[i]);
  printf("hipHostMalloc 4 KB %.1f us, hipHostFree %.1f us\n", a / N, (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) (void)hipHostMalloc(&p[i], 4096, hipHostMallocMapped);
  a = now() - t; t = now();
  for (int i = 0; i < N; ++i) (void)hipHostFree(p[i]);
  printf("hipHostMalloc mapped 4 KB %.1f us, hipHostFree %.1f us\n", a / N, (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) (void)hipMalloc(&p[i], 65536);
  a = now() - t; t = now();
  for (int i = 0; i < N; ++i) (void)hipFree(p[i]);
  printf("hipMalloc 64 KB %.1f us, hipFree %.1f us\n", a / N, (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) (void)hipMalloc(&p[i], 4 << 20);
  a = now() - t; t = now();
  for (int i = 0; i < N; ++i) (void)hipFree(p[i]);
  printf("hipMalloc 4 MB %.1f us, hipFree %.1f us\n", a / N, (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
  a = now() - t; t = now();
  for (int i = 0; i < N; ++i) (void)hipStreamDestroy(st[i]);
  printf("hipStreamCreate %.1f us, hipStreamDestroy %.1f us\n", a / N, (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) (void)hipEventCreate(&ev[i]);
  a = now() - t; t = now();
  for (int i = 0; i < N; ++i) (void)hipEventDestroy(ev[i]);
  printf("hipEventCreate %.1f us, hipEventDestroy %.1f us\n", a / N, (now() - t) / N);
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  void *d; (void)hipMalloc(&d, 1 << 20);
  t = now();
  for (int i = 0; i < N; ++i) (void)hipMemsetAsync(d, 0, 4096, s);
  (void)hipStreamSynchronize(s);
  printf("hipMemsetAsync 4 KB %.1f us (enqueue + run, back to back)\n", (now() - t) / N);
  char hbuf[4096];
  t = now();
  for (int i = 0; i < N; ++i) { (void)hipMemcpyAsync(d, hbuf, 4096, hipMemcpyHostToDevice, s); }
  (void)hipStreamSynchronize(s);
  printf("hipMemcpyAsync H2D 4 KB pageable %.1f us\n", (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) { (void)hipMemcpyAsync(hbuf, d, 4096, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }
  printf("hipMemcpyAsync D2H 4 KB pageable + sync %.1f us\n", (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) (void)hipStreamSynchronize(s);
  printf("hipStreamSynchronize (idle) %.1f us\n", (now() - t) / N);
  return 0;
}
