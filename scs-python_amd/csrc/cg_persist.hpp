// cg_persist.hpp — the whole warm-started PCG solve of one ADMM iteration in ONE launch (small problems).
//
// Plays the role of the CG loop of scs_source/linsys/cpu/indirect/private.c (R:meson.build:261; absent;
// algorithm: SURVEY App. A.4).  For problems whose mat-vec takes a few microseconds the launch-per-kernel
// path (work_admm.inl: enqueue_lin_sys_head + 4 launches per CG step) is all launch latency: ~5-6 us per
// kernel on this GPU versus < 1 us of work.  Here a small persistent grid runs
//     tolerance -> y0 = v_y + R_y^{-1} A ws -> r0, p0 -> { z = R_y^{-1} A p -> Gp, p'Gp -> x, r, y updates -> beta, p }*
// with a grid barrier between the phases and no host round trip.  Every phase executes the SAME per-block
// bodies (spmv_stream_block, cg_update_block) over the SAME virtual block decomposition and reduces the same
// partial arrays in the same order as the launch-per-kernel path, so both paths give identical bits
// (tests/test_hip_parity.py::test_persistent_cg_bit_identical).
//
// Grid: gridDim.x workgroups x NG groups of 256 lanes.  gridDim.x == 1 needs no inter-workgroup barrier at
// all; gridDim.x > 1 uses a monotonic atomic counter (agent-scope release/acquire = L2 write-back/invalidate
// across XCDs, ~1-2 us for <= 32 workgroups: tools/gridbar_bench.hip) and the host launches at most
// kCgPersistMaxWgs workgroups so that co-residency is never in question; a spin budget turns a would-be hang
// into an error flag.
#pragma once
#include "spmv.hpp"
#include "vec.hpp"

namespace scship {

constexpr int kCgPersistMaxWgs = 16;
template <int NG>
constexpr size_t cg_persist_lds() { return (size_t)(NG * kNnzPerWg + NG * (kVecThreads / 64) + 4) * sizeof(double); }

struct CgPersistArgs {
  CsrView Ar, At, Pf;
  int has_P, n, m;
  const double *diag_r, *v, *ws;  // ws: warm start x0 (length n)
  double *ut;                     // x in ut[0:n) (already = ws), y in ut[n:n+m)
  double *r, *p, *Gp, *z;         // CG work vectors (n, n, n, m)
  const double *M;                // Jacobi preconditioner (n)
  double *part, *part2;           // reduction partials (SpMV epilogues / vector kernels)
  const double *part_p;           // k_prep's partials: [max |ws| (np_p) | max |rhs| (np_p)]
  int np_p;
  const double *params;           // P_RES_MIN, P_IPOW
  double *sc;
  int *fl;
  int max_its;
  unsigned *bar;                  // [0] grid barrier counter, [1] exit counter; both 0 at entry and at exit
};

template <int NG>
__global__ __launch_bounds__(kVecThreads *NG) void k_cg_persist(CgPersistArgs a) {
  static_assert(kVecThreads == kSpmvThreads, "one group = one virtual workgroup of either kernel family");
  extern __shared__ __attribute__((aligned(16))) double smem[];  // prod[NG][kNnzPerWg] | red[NG][4] | bcast[4]
  double(*prod)[kNnzPerWg] = reinterpret_cast<double(*)[kNnzPerWg]>(smem);
  double(*red)[kVecThreads / 64] = reinterpret_cast<double(*)[kVecThreads / 64]>(smem + NG * kNnzPerWg);
  double *bcast = smem + NG * kNnzPerWg + NG * (kVecThreads / 64);
  const int g = threadIdx.x / kVecThreads, tid = threadIdx.x % kVecThreads;
  const int G = gridDim.x * NG, me = blockIdx.x * NG + g;
  const BlockSync sync{};
  unsigned bar_target = 0;
  auto gbar = [&]() {
    if (gridDim.x == 1) {
      __syncthreads();
      return;
    }
    bar_target += gridDim.x;
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_fetch_add(a.bar, 1u, __ATOMIC_RELEASE);
      long spins = 0;
      while (__atomic_load_n(a.bar, __ATOMIC_ACQUIRE) < bar_target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1L << 26)) { a.fl[F_PERSIST_ERR] = 1; break; }  // ~seconds: some workgroup never arrived
      }
    }
    __syncthreads();
  };
  // reduce a partial array exactly like the single-workgroup finalize kernels do, result to all lanes
  auto all_sum = [&](const double *part, int np) {
    const double t = part_sum(part, np, red[g], tid, sync);
    if (threadIdx.x == 0) bcast[0] = t;
    __syncthreads();
    const double r = bcast[0];
    __syncthreads();
    return r;
  };
  auto all_max = [&](const double *part, int np) {
    const double t = part_max(part, np, red[g], tid, sync);
    if (threadIdx.x == 0) bcast[0] = t;
    __syncthreads();
    const double r = bcast[0];
    __syncthreads();
    return r;
  };
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  const int n = a.n, m = a.m;
  const int nbA = a.Ar.nblk, nbAt = a.At.nblk, nbP = a.has_P ? a.Pf.nblk : 0;
  const int itA = (nbA + G - 1) / G, itAt = (nbAt + G - 1) / G, itP = (nbP + G - 1) / G;

  // ---- tolerance and the zero right-hand-side short circuit (k_fin_tol) ----
  const double ws_norm = all_max(a.part_p, a.np_p);
  const double rhs_norm = all_max(a.part_p + a.np_p, a.np_p);
  const double tol = fmax(1e-12, 0.2 * fmin(a.params[P_RES_MIN], ws_norm / a.params[P_IPOW]));
  if (lead) {
    a.sc[S_TOL] = tol;
    a.sc[S_WSNORM] = ws_norm;
    a.fl[F_ZERO_RHS] = rhs_norm <= 1e-12 ? 1 : 0;
  }
  if (rhs_norm <= 1e-12) {  // uniform over the grid
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n + m; i += (long)gridDim.x * blockDim.x) a.ut[i] = 0.;
    if (lead) { a.fl[F_ITERS] = 0; a.fl[F_DONE] = 1; }
    return;
  }

  // ---- fused, warm-started CG start (EpiY, EpiR0; see k_prep) ----
  for (int it = 0; it < itA; ++it) {
    const int b = me + it * G;
    spmv_stream_block<EpiY, true>(a.Ar, a.ws, EpiY{a.ut + n, a.diag_r + n, a.v + n}, b, nbA, prod[g], red[g], tid, sync, b < nbA);
  }
  for (int it = 0; it < itP; ++it) {
    const int b = me + it * G;
    spmv_stream_block<EpiStore, true>(a.Pf, a.ws, EpiStore{a.Gp, 0}, b, nbP, prod[g], red[g], tid, sync, b < nbP);
  }
  gbar();
  for (int it = 0; it < itAt; ++it) {
    const int b = me + it * G;
    spmv_stream_block<EpiR0, true>(a.At, a.ut + n, EpiR0{a.r, a.p, a.M, a.diag_r, a.v, a.ws, a.has_P ? a.Gp : nullptr, a.part}, b, nbAt,
                                   prod[g], red[g], tid, sync, b < nbAt);
  }
  gbar();
  double ztr = all_sum(a.part, nbAt);               // k_fin_cg_init, partial layout [sum | max]
  double rn = all_max(a.part + nbAt, nbAt);
  int iters = 0;
  bool done = rn < fmax(tol, 1e-12);
  double alpha = 0., beta = 0.;
  const int nbu = vec_blocks(n > m ? n : m), nbd = vec_blocks(n);
  const int itU = (nbu + G - 1) / G;

  while (!done && iters < a.max_its) {
    // z = R_y^{-1} A p  (and P p)
    for (int it = 0; it < itA; ++it) {
      const int b = me + it * G;
      spmv_stream_block<EpiDivR, true>(a.Ar, a.p, EpiDivR{a.z, a.diag_r + n}, b, nbA, prod[g], red[g], tid, sync, b < nbA);
    }
    for (int it = 0; it < itP; ++it) {
      const int b = me + it * G;
      spmv_stream_block<EpiStore, true>(a.Pf, a.p, EpiStore{a.Gp, 0}, b, nbP, prod[g], red[g], tid, sync, b < nbP);
    }
    gbar();
    // Gp = A'z + R_x p (+ P p), partial p'Gp
    for (int it = 0; it < itAt; ++it) {
      const int b = me + it * G;
      spmv_stream_block<EpiGp, true>(a.At, a.z, EpiGp{a.Gp, a.p, a.diag_r, a.has_P ? 1 : 0, a.part}, b, nbAt, prod[g], red[g], tid,
                                     sync, b < nbAt);
    }
    gbar();
    alpha = ztr / all_sum(a.part, nbAt);  // k_cg_update prologue
    for (int it = 0; it < itU; ++it) {
      const int b = me + it * G;
      cg_update_block(a.ut, a.r, a.p, a.Gp, a.M, n, a.ut + n, a.z, m, [&]() { return alpha; }, a.part2, b, nbu, tid, red[g], sync, b < nbu);
    }
    gbar();
    rn = all_max(a.part2, nbu);  // k_cg_dir prologue
    const double ztr_new = all_sum(a.part2 + nbu, nbu);
    beta = ztr_new / ztr;
    ztr = ztr_new;
    ++iters;
    done = rn < tol;
    if (!done) {
      for (int b = me; b < nbd; b += G)
        for (long i = (long)b * kVecThreads + tid; i < n; i += (long)nbd * kVecThreads) a.p[i] = a.M[i] * a.r[i] + beta * a.p[i];
    }
    gbar();  // p complete; `part` may be overwritten again
  }
  if (lead) {
    a.sc[S_RNORM] = rn;
    a.sc[S_ALPHA] = alpha;
    a.sc[S_BETA] = beta;
    a.fl[F_ITERS] = iters;
    a.fl[F_DONE] = 1;
  }
  // the workgroup that leaves last (everybody is past the final barrier by then) re-arms the counters
  if (gridDim.x > 1 && threadIdx.x == 0 && __atomic_fetch_add(a.bar + 1, 1u, __ATOMIC_ACQ_REL) == gridDim.x - 1) {
    a.bar[0] = 0;
    a.bar[1] = 0;
  }
}

}  // namespace scship
