// psd_lab.hip — phase breakdown of the batched PSD projection kernel (K9) with PSD_PROFILE timers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPSD_PROFILE=1 -o devtools/psd_lab tools/psd_lab.hip
//   ./devtools/psd_lab [order] [count] [calls] [perturbation] [split: 0 one launch, 1 split mode, G >= 2 split mode with the sweeps
//                       of each matrix over G workgroups (k_psd_sweep_mc)] [compare: 1 = rerun in split mode 1 and compare the bits]
// First call is cold (V = I); later calls are warm-started on a matrix perturbed by `perturbation` (relative),
// which is what consecutive ADMM iterations look like.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../scs-python_amd/csrc/psd.hpp"

using namespace scship;

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 200, cnt = argc > 2 ? atoi(argv[2]) : 50, calls = argc > 3 ? atoi(argv[3]) : 6;
  const double pert = argc > 4 ? atof(argv[4]) : 1e-3;
  const int split_arg = argc > 5 ? atoi(argv[5]) : 0;
  const bool compare = argc > 6 && atoi(argv[6]) != 0;
  int *d_err;
  HIP_CHECK(hipMalloc(&d_err, 4));
  HIP_CHECK(hipMemset(d_err, 0, 4));
  std::vector<std::vector<double>> outs[2];
  for (int pass = 0; pass < (compare ? 2 : 1); ++pass) {
  const int mc = pass == 0 ? split_arg : 1;
  const bool split = mc != 0;
  const long vlen = (long)n * (n + 1) / 2;
  std::mt19937_64 g(1);
  std::normal_distribution<double> nd;
  std::vector<double> x0(cnt * vlen), x(cnt * vlen);
  for (auto &v : x0) v = nd(g);
  std::vector<int> off(cnt), ord(cnt, n);
  std::vector<long> woff(cnt);
  long wtot = 0;
  for (int c = 0; c < cnt; ++c) { off[c] = (int)(c * vlen); woff[c] = wtot; wtot += psd_scratch_doubles(n); }
  int *d_off, *d_ord; long *d_woff; double *d_x, *d_scr;
  HIP_CHECK(hipMalloc(&d_off, cnt * 4)); HIP_CHECK(hipMalloc(&d_ord, cnt * 4)); HIP_CHECK(hipMalloc(&d_woff, cnt * 8));
  HIP_CHECK(hipMalloc(&d_x, x.size() * 8)); HIP_CHECK(hipMalloc(&d_scr, wtot * 8));
  HIP_CHECK(hipMemcpy(d_off, off.data(), cnt * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_ord, ord.data(), cnt * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_woff, woff.data(), cnt * 8, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemset(d_scr, 0, wtot * 8));
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_psd<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPsdLdsBytes));
  PsdBatch B{d_off, d_ord, d_woff, cnt};
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  std::printf("order %d (padded %ld), %d matrices, perturbation %.1e; ticks are 10 ns (thread 0 of matrix 0)\n", n, psd_np(n), cnt, pert);
  std::printf("call   total_us | unpack  warmGEMM   pivots  updates  norms+sched  reconstruct | sweeps  steps  pivot_us/step  update_us/step\n");
  const long np = psd_np(n);
  const long st_off = psd_scratch_doubles(n) - kPsdStateDoubles;  // state[] at the end of matrix 0's scratch
  for (int call = 0; call < calls; ++call) {
    for (size_t i = 0; i < x.size(); ++i) x[i] = x0[i] * (1.0 + pert * call) + pert * call * nd(g);
    HIP_CHECK(hipMemcpy(d_x, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipEventRecord(e0));
    if (n <= kPsdSmallMax) {  // four wavefronts per matrix; PSD_LAB_SMALL1 (and the second pass of `compare`): the one-wavefront kernel
      if (getenv("PSD_LAB_SMALL1") || (compare && pass == 1))
        hipLaunchKernelGGL(k_proj_psd_small, dim3(cnt), dim3(64), 0, 0, d_x, B, d_scr, 1, (const int *)nullptr, (const double *)nullptr);
      else
        hipLaunchKernelGGL(k_proj_psd_small4, dim3(cnt), dim3(kPsdSmallThreads), 0, 0, d_x, B, d_scr, 1, (const int *)nullptr, (const double *)nullptr);
    }
    else if (split) {  // the launch sequence of scs_hip.hip launch_psd (round 5: 1-D XCD-aware grids, refinement stage; PSD_LAB_REFINE0: strict sweeps)
      const int ntile = (int)np / 16;
      const int gper = psd_gemm_tasks(ntile);
      const dim3 gg(psd_gemm_grid(gper, cnt)), gb(kPsdGemmThreads), gt(psd_xcd_grid(ntile, cnt));
      const PsdRefineCfg R = psd_refine_default(!getenv("PSD_LAB_REFINE0"));
      hipLaunchKernelGGL(k_psd_front, gt, dim3(kPsdFrontThreads), 0, 0, (const double *)d_x, B, d_scr, 1, (const int *)nullptr);
      hipLaunchKernelGGL(k_proj_psd<3>, dim3(cnt), dim3(kPsdThreads), kPsdLdsBytes, 0, d_x, B, d_scr, 1, 0, (const int *)nullptr, (const double *)nullptr,
                         psd_refine_default(false), 0);
      hipLaunchKernelGGL(k_psd_gemm<PSD_G1>, gg, gb, 0, 0, d_x, B, d_scr, 1, nullptr, gper);
      hipLaunchKernelGGL(k_psd_gemm<PSD_G2>, gg, gb, 0, 0, d_x, B, d_scr, 1, nullptr, gper);
      for (int round = 0; round < 2; ++round) {
        int post = (R.on && round == 1) ? 1 : 0;
        if (post) {
          hipLaunchKernelGGL(k_psd_gemm<PSD_COMM>, gg, gb, 0, 0, d_x, B, d_scr, 1, nullptr, gper);
          hipLaunchKernelGGL(k_psd_gemm<PSD_KK>, gg, gb, 0, 0, d_x, B, d_scr, 1, nullptr, gper);
          hipLaunchKernelGGL(k_psd_gemm<PSD_T>, gg, gb, 0, 0, d_x, B, d_scr, 1, nullptr, gper);
          hipLaunchKernelGGL(k_psd_gemm<PSD_S1>, gg, gb, 0, 0, d_x, B, d_scr, 1, nullptr, gper);
          hipLaunchKernelGGL(k_psd_apply_q, gt, dim3(kPsdApplyThreads), (size_t)32 * np * 8, 0, B, d_scr, nullptr);
        }
        if (mc >= 2) {
          int G = mc, rnd = round;
          const int *st = nullptr;
          int la = getenv("PSD_LAB_LA0") ? 0 : 1;  // look-ahead (one barrier per step)
          const double *tl = nullptr;
          PsdRefineCfg Rr = R;
          long budget = 1L << 25;
          void *args[] = {&B, &d_scr, &rnd, &G, &la, &d_err, &st, &tl, &Rr, &post, &budget};
          if (getenv("PSD_LAB_PLAIN"))  // ordinary launch (e.g. under rocprofv3)
            hipLaunchKernelGGL(k_psd_sweep_mc, dim3((unsigned)psd_mc_grid(cnt, G)), dim3(kPsdThreads), kPsdMcLdsBytes, 0, B, d_scr, rnd, G, la, d_err, st, tl, Rr,
                               post, budget);
          else
          HIP_CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_psd_sweep_mc), dim3((unsigned)psd_mc_grid(cnt, G)), dim3(kPsdThreads),
                                               args, (unsigned)kPsdMcLdsBytes, 0));
        } else
        hipLaunchKernelGGL(k_proj_psd<1>, dim3(cnt), dim3(kPsdThreads), kPsdLdsBytes, 0, d_x, B, d_scr, 1, round, (const int *)nullptr, (const double *)nullptr,
                           R, post);
        hipLaunchKernelGGL(k_psd_apply_v, gt, dim3(kPsdApplyThreads), (size_t)16 * np * 8, 0, B, d_scr, nullptr);
      }
      hipLaunchKernelGGL(k_psd_fmap, gt, dim3(256), 0, 0, B, d_scr, (const int *)nullptr);
      hipLaunchKernelGGL(k_psd_gemm<PSD_R1>, gg, gb, 0, 0, d_x, B, d_scr, 1, nullptr, gper);
      hipLaunchKernelGGL(k_psd_gemm<PSD_R2>, gg, gb, 0, 0, d_x, B, d_scr, 1, nullptr, gper);
    } else hipLaunchKernelGGL(k_proj_psd<0>, dim3(cnt), dim3(kPsdThreads), kPsdLdsBytes, 0, d_x, B, d_scr, 1, 0, (const int *)nullptr, (const double *)nullptr,
                              psd_refine_default(false), 0);
    HIP_CHECK(hipEventRecord(e1)); HIP_CHECK(hipEventSynchronize(e1));
    float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    double st[8];
    HIP_CHECK(hipMemcpy(st, d_scr + st_off, sizeof st, hipMemcpyDeviceToHost));
    outs[pass].emplace_back(x.size());
    HIP_CHECK(hipMemcpy(outs[pass].back().data(), d_x, x.size() * 8, hipMemcpyDeviceToHost));
#if PSD_PROFILE
    {
      double ps[8];
      HIP_CHECK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(psd_prof_stop), sizeof ps));
      std::printf("     last stopping test of workgroup 0 (mode %.0f, %.0f tests so far): diagonal %.2f us, reciprocals %.2f, sums%s %.2f, reductions + decision %.2f\n",
                  ps[5], ps[4], ps[0] / 100, ps[1] / 100, ps[5] == 1. ? " + K1" : "", ps[2] / 100, ps[3] / 100);
    }
#endif
    const double steps = n <= kPsdSmallMax ? st[7] * (((n + 1) & ~1) - 1) : st[7] * (np / 8 - 1);  // rounds of the one-wave kernel / outer steps
    std::printf("%3d  %9.1f | %6.1f  %8.1f  %7.1f  %7.1f  %11.1f  %11.1f | %6.0f  %5.0f  %13.2f  %14.2f\n", call, ms * 1e3, st[1] / 100, st[2] / 100,
                st[3] / 100, st[4] / 100, st[5] / 100, st[6] / 100, st[7], steps, steps ? st[3] / 100 / steps : 0., steps ? st[4] / 100 / steps : 0.);
  }
  int herr = 0;
  HIP_CHECK(hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost));
  if (herr) std::printf("BARRIER TIMEOUT flagged\n");
  {  // orthogonality of the warm-start basis of matrix 0 after all calls
    const bool small = n <= kPsdSmallMax;
    const long N = small ? ((n + 1) & ~1) : np;
    std::vector<double> Vh(N * N);
    HIP_CHECK(hipMemcpy(Vh.data(), d_scr + (small ? 0 : np * np), Vh.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.;
    for (long i = 0; i < N; ++i)
      for (long j = 0; j <= i; ++j) {
        double acc = 0.;
        for (long k = 0; k < N; ++k) acc += Vh[k + N * i] * Vh[k + N * j];
        worst = std::max(worst, std::fabs(acc - (i == j ? 1. : 0.)));
      }
    std::printf("max |V'V - I| after %d calls: %.3e\n", calls, worst);
  }
  g.seed(1);
  nd.reset();
  for (auto &v : x0) v = nd(g);  // same inputs in the second pass
  }  // pass
  if (compare) {
    long diff = 0;
    for (size_t c = 0; c < outs[0].size(); ++c)
      for (size_t i = 0; i < outs[0][c].size(); ++i) diff += std::memcmp(&outs[0][c][i], &outs[1][c][i], 8) != 0;
    double worst = 0., scale = 0.;
    for (size_t c = 0; c < outs[0].size(); ++c)
      for (size_t i = 0; i < outs[0][c].size(); ++i) {
        worst = std::max(worst, std::fabs(outs[0][c][i] - outs[1][c][i]));
        scale = std::max(scale, std::fabs(outs[1][c][i]));
      }
    std::printf("bitwise differences between the two passes over %zu calls: %ld; max |difference| %.3e (entries up to %.3e)\n", outs[0].size(), diff,
                worst, scale);
  }
  return 0;
}
