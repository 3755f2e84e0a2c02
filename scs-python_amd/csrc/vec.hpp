// vec.hpp — K4: fused vector kernels of the PCG and of the ADMM step, with
// fixed-order two-stage reductions and all loop scalars kept on the device.
//
// Plays the role of scs_source/src/linalg.c (named at R:meson.build:191; absent)
// and replaces the cuBLAS level-1 calls of GPU_INDIRECT (R:legacy_setup.py:263).
// Every kernel is one pass over its vectors (HBM-bound).  Reductions are two-stage
// and fixed-order: a kernel leaves per-block partials, and the CONSUMER kernel
// re-reduces them in its prologue (every workgroup, same order => same scalar
// everywhere): alpha, beta, tau, the iterate norm and the CG tolerance never
// cost a separate launch, and never a host round trip (SURVEY §7 "hard parts").
// Single-workgroup finalize kernels remain only off the per-iteration path
// (cold KKT solves at init / scale updates, residual checks, AA).
#pragma once
#include "common.hpp"

namespace scship {

constexpr int kVecThreads = 256;
constexpr int kMaxVecBlocks = 2048;
// Elements per lane a vector kernel's grid is sized for: 4, and 8 for vectors beyond 2^20 elements (round 4).  The consumer of a
// reduction re-reduces ALL partials in the prologue of every workgroup (no finalize launch): at m = 2e6, k_cg_dir's 977 workgroups each read
// the 2 x 1954 partials k_cg_update left — as many bytes from L2 as the kernel's payload.  Halving the grid of the long vectors: +1.2 % on
// the metric window (308 -> 312 iters/s); 8 everywhere costs config 2 (m = 2e5) 2.7 %, 16 costs the metric workload 2 % (tools/dbg/r4_ve.sh).
constexpr long kVecLong = 1L << 20;

// device scalar slots (double)
enum : int {
  S_ZTR = 0, S_ZTR_B, S_ALPHA, S_BETA, S_TOL, S_RNORM, S_TAUT, S_VSCALE, S_GG, S_WSNORM, S_AA_NORMG, S_AA_NORMD,
  S_AA_NORM, S_AA_REG, S_BOX_T, S_BOX_STOP, S_TMP1, S_TMP2, S_TMP3, S_COUNT = 32
};
// device flag slots (int)
enum : int {
  F_DONE = 0, F_ITERS, F_AA_SUCCESS, F_AA_ITER, F_AA_ACCEPT, F_AA_REJ_LAPACK, F_AA_REJ_RANK0, F_AA_REJ_NONFINITE,
  F_AA_REJ_WEIGHT, F_AA_SAFE_REJ, F_AA_LAST_RANK, F_AA_CALLS, F_SAFE_OK, F_SAFE_BAD, F_ZERO_RHS, F_STEP, F_PERSIST_ERR, F_STALL, F_COUNT = 32
};

// F_STALL (run-ahead mode of the ADMM loop, loop.hpp / work_admm.inl): the host enqueues a whole iteration — head, a chunk of
// CG steps, tau / cones / v update — and the next one before it looks at the CG flags.  If the chunk was too short
// (F_DONE still 0 when the first kernel after it runs) that kernel raises F_STALL and F_DONE, and every kernel
// queued behind it returns at once (the CG-step kernels through F_DONE, the others through F_STALL), so nothing is
// computed from an unconverged linear solve; the host then lowers both flags and resumes from the intact CG state.

// per-iteration host scalars, kept in mapped pinned memory so that captured hipGraphs stay static
enum : int { P_DO_SCALE = 0, P_RES_MIN, P_IPOW, P_FIRST, P_PSD_TOL2, P_COUNT = 8 };

__host__ __device__ inline int vec_blocks(long n) {
  // (round 4 A/B on the metric workload: 8 per lane already from 2^19, or 16 per lane beyond 2^20 / 2^21, are 1-2 % slower)
  const long per = (n > kVecLong ? 8L : 4L) * kVecThreads;
  long nb = (n + per - 1) / per;
  if (nb < 1) nb = 1;
  if (nb > kMaxVecBlocks) nb = kMaxVecBlocks;
  return (int)nb;
}

// ---- single-workgroup reduction of a partial array (fixed order) -------------
// (group form: kVecThreads consecutive lanes, tid = index inside the group — see common.hpp)
template <class Sync>
__device__ __forceinline__ double part_sum(const double *part, int np, double *sm, int tid, Sync sync) {
  double s = 0.;
  for (int i = tid; i < np; i += kVecThreads) s += part[i];
  return group_sum<kVecThreads>(s, sm, tid, sync);
}
template <class Sync>
__device__ __forceinline__ double part_max(const double *part, int np, double *sm, int tid, Sync sync) {
  double s = 0.;
  for (int i = tid; i < np; i += kVecThreads) s = fmax(s, part[i]);
  return group_max<kVecThreads>(s, sm, tid, sync);
}
__device__ __forceinline__ double part_sum(const double *part, int np, double *sm) {
  return part_sum(part, np, sm, (int)threadIdx.x, BlockSync{});
}
__device__ __forceinline__ double part_max(const double *part, int np, double *sm) {
  return part_max(part, np, sm, (int)threadIdx.x, BlockSync{});
}

// ||v||_2 partials
__device__ __forceinline__ void d_sumsq(const double *__restrict__ v, long n, double *part) {
  __shared__ double sm[kVecThreads / 64];
  double s = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) s += v[i] * v[i];
  s = block_sum<kVecThreads>(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(kVecThreads) void k_sumsq(const double *__restrict__ v, long n, double *part) {
  d_sumsq(v, n, part);
}

// Start of project_lin_sys: normalise v, snapshot v_prev and form the CG warm start
//   ws = u_x + tau g_x   (also copied into ut_x: the CG iterate x starts there).
// The KKT right-hand side [R_x v_x; -R_y v_y] is never materialised: the warm-started CG only needs
//   r0 = rhs_x + A' R_y^{-1} rhs_y - (R_x + P + A' R_y^{-1} A) ws = R_x (v_x - ws) - P ws - A' (v_y + R_y^{-1} A ws)
// which costs ONE A product and ONE A' product (epilogues EpiY / EpiR0) instead of three.
// Partials: [max |ws| , max |rhs|] (the latter for the zero-rhs short-circuit).
// The iterate normalisation factor sqrt(l) / ||v|| is formed in the prologue from the
// sum-of-squares partials `vpart` (written by k_v_update of the previous iteration, or by k_sumsq after
// anything else touched v): every workgroup reduces them in the same fixed order, no finalize launch.
__device__ __forceinline__ void d_prep(double *v, double *v_prev, double *ut, double *ws,
                                       const double *__restrict__ u, const double *__restrict__ g,
                                       const double *__restrict__ diag_r, int n, int m, const double *params,
                                       const double *vpart, int nvp, double *sc, double *part,
                                       const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc;
  const long l = (long)n + m + 1;
  {
    const double ss = part_sum(vpart, nvp, sm);
    if (threadIdx.x == 0) {
      bc = sqrt((double)l) / fmax(sqrt(ss), 1e-300);
      if (blockIdx.x == 0) sc[S_VSCALE] = bc;
    }
    __syncthreads();
  }
  const double scale = params[P_DO_SCALE] != 0. ? bc : 1.0;
  const double tau = u[l - 1];
  double mx = 0., mr = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < l; i += (long)gridDim.x * kVecThreads) {
    const double vi = v[i] * scale;
    v[i] = vi;
    v_prev[i] = vi;
    if (i < n) {
      const double w = u[i] + tau * g[i];
      ws[i] = w;
      ut[i] = w;
      mx = fmax(mx, abs_nan_inf(w));
      mr = fmax(mr, abs_nan_inf(diag_r[i] * vi));
    } else if (i < l - 1) {
      mr = fmax(mr, abs_nan_inf(diag_r[i] * vi));
    }
  }
  mx = block_max<kVecThreads>(mx, sm);
  mr = block_max<kVecThreads>(mr, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = mx;
    part[gridDim.x + blockIdx.x] = mr;  // ||rhs||_inf of the KKT system
  }
}
__global__ __launch_bounds__(kVecThreads) void k_prep(double *v, double *v_prev, double *ut, double *ws,
                                                      const double *__restrict__ u,
                                                      const double *__restrict__ g,
                                                      const double *__restrict__ diag_r, int n, int m,
                                                      const double *params, const double *vpart, int nvp,
                                                      double *sc, double *part, const int *stall) {
  d_prep(v, v_prev, ut, ws, u, g, diag_r, n, m, params, vpart, nvp, sc, part, stall);
}

// CG tolerance: tol = max(1e-12, 0.2 * min(res_min, ||ws||_inf / (k+1)^1.5))
// A right-hand side with ||rhs||_inf <= 1e-12 short-circuits to the zero solution (F_ZERO_RHS).
__device__ __forceinline__ void d_fin_tol(const double *part, int np, double res_min, double ipow,
                                          double fixed_tol, int have_rhs_norm, const double *params,
                                          double *sc, int *fl) {
  __shared__ double sm[kVecThreads / 64];
  if (params) {
    res_min = params[P_RES_MIN];
    ipow = params[P_IPOW];
  }
  const double ws = part_max(part, np, sm);
  const double rn = have_rhs_norm ? part_max(part + np, np, sm) : 1.0;
  if (threadIdx.x == 0) {
    double tol = fixed_tol;
    if (fixed_tol <= 0.) {
      tol = 0.2 * fmin(res_min, ws / ipow);
      tol = fmax(1e-12, tol);
    }
    sc[S_TOL] = tol;
    sc[S_WSNORM] = ws;
    const int zero = (rn <= 1e-12) ? 1 : 0;
    fl[F_ZERO_RHS] = zero;
    fl[F_DONE] = zero;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_fin_tol(const double *part, int np, double res_min,
                                                         double ipow, double fixed_tol, int have_rhs_norm,
                                                         const double *params, double *sc, int *fl) {
  d_fin_tol(part, np, res_min, ipow, fixed_tol, have_rhs_norm, params, sc, fl);
}

// r = b - G ws; x = ws; p = M r; partial [max|r| , sum r M r]
// Gws2 != nullptr: G ws = Gws + Gws2 (split layout of A', see EpiGp::split)
__device__ __forceinline__ void d_cg_init(const double *__restrict__ b, const double *__restrict__ Gws,
                                          const double *__restrict__ ws, const double *__restrict__ M,
                                          double *x, double *r, double *p, int n, int have_ws, const int *fl,
                                          double *part, const double *__restrict__ Gws2) {
  __shared__ double sm[kVecThreads / 64];
  double mx = 0., s = 0.;
  if (fl[F_ZERO_RHS]) {  // zero right-hand side: the solution is zero, no iterations
    for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) x[i] = 0.;
    if (threadIdx.x == 0) {
      part[blockIdx.x] = 0.;
      part[gridDim.x + blockIdx.x] = 0.;
    }
    return;
  }
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) {
    const double ri = have_ws ? b[i] - (Gws2 ? Gws[i] + Gws2[i] : Gws[i]) : b[i];
    const double zi = M[i] * ri;
    x[i] = have_ws ? ws[i] : 0.0;
    r[i] = ri;
    p[i] = zi;
    mx = fmax(mx, abs_nan_inf(ri));
    s += zi * ri;
  }
  mx = block_max<kVecThreads>(mx, sm);
  s = block_sum<kVecThreads>(s, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = mx;
    part[gridDim.x + blockIdx.x] = s;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_cg_init(const double *__restrict__ b,
                                                         const double *__restrict__ Gws,
                                                         const double *__restrict__ ws,
                                                         const double *__restrict__ M, double *x, double *r,
                                                         double *p, int n, int have_ws, const int *fl,
                                                         double *part,
                                                         const double *__restrict__ Gws2 = nullptr) {
  d_cg_init(b, Gws, ws, M, x, r, p, n, have_ws, fl, part, Gws2);
}

// sum_first: partial layout [sum | max] (SpMV epilogue) instead of [max | sum] (k_cg_init)
__device__ __forceinline__ void d_fin_cg_init(const double *part, int np, int sum_first, double *sc, int *fl) {
  __shared__ double sm[kVecThreads / 64];
  const double rn = part_max(sum_first ? part + np : part, np, sm);
  const double ztr = part_sum(sum_first ? part : part + np, np, sm);
  if (threadIdx.x == 0) {
    sc[S_RNORM] = rn;
    sc[((fl[F_STEP] + 1) & 1) ? S_ZTR_B : S_ZTR] = ztr;  // the first CG step bumps F_STEP, then reads this slot
    if (rn < fmax(sc[S_TOL], 1e-12)) fl[F_DONE] = 1;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_fin_cg_init(const double *part, int np, int sum_first,
                                                             double *sc, int *fl) {
  d_fin_cg_init(part, np, sum_first, sc, fl);
}

// ADMM path: k_fin_tol + k_fin_cg_init + the CG step counter reset + the zero-rhs short circuit in ONE
// single-workgroup launch after the fused CG start (the two SpMVs of the start do not depend on the tolerance).
//   prep_part = k_prep's [max |ws| (np_p) | max |rhs| (np_p)],  r0_part = EpiR0's [sum r0'M r0 (np_r) | max |r0| (np_r)]
__device__ __forceinline__ void d_fin_head(const double *prep_part, int np_p, const double *r0_part, int np_r,
                                           const double *params, double *sc, int *fl, double *ut, long nm,
                                           const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double sm[kVecThreads / 64];
  __shared__ int zero_rhs;
  const double ws = part_max(prep_part, np_p, sm);
  const double rhs = part_max(prep_part + np_p, np_p, sm);
  const double ztr = part_sum(r0_part, np_r, sm);
  const double rn = part_max(r0_part + np_r, np_r, sm);
  if (threadIdx.x == 0) {
    const double tol = fmax(1e-12, 0.2 * fmin(params[P_RES_MIN], ws / params[P_IPOW]));
    const int zero = (rhs <= 1e-12) ? 1 : 0;
    sc[S_TOL] = tol;
    sc[S_WSNORM] = ws;
    sc[S_RNORM] = rn;
    sc[((fl[F_STEP] + 1) & 1) ? S_ZTR_B : S_ZTR] = ztr;  // the first CG step bumps F_STEP, then reads this slot
    fl[F_ZERO_RHS] = zero;
    fl[F_ITERS] = 0;
    fl[F_DONE] = (zero || rn < fmax(tol, 1e-12)) ? 1 : 0;
    zero_rhs = zero;
  }
  __syncthreads();
  if (zero_rhs)  // zero right-hand side => zero solution (happens at most at the first iteration of a cold start)
    for (long i = threadIdx.x; i < nm; i += kVecThreads) ut[i] = 0.;
}
__global__ __launch_bounds__(kVecThreads) void k_fin_head(const double *prep_part, int np_p,
                                                          const double *r0_part, int np_r,
                                                          const double *params, double *sc, int *fl,
                                                          double *ut, long nm, const int *stall) {
  d_fin_head(prep_part, np_p, r0_part, np_r, params, sc, fl, ut, nm, stall);
}

// x += alpha p; r -= alpha Gp; partial [max|r|, sum r M r].
// With yacc != nullptr also y += alpha z (z = R_y^{-1} A p of this step): the y-block of the KKT solution
// y = R_y^{-1}(A x - r_y) is carried along the CG recurrence instead of a final A x product.
// alpha = z'r / p'Gp is formed in the prologue: every workgroup reduces the p'Gp partials of the A' kernel
// itself (a few hundred L2-resident doubles, fixed order => identical alpha everywhere), so no separate
// single-workgroup "finalize" launch sits between the SpMV and this kernel.  z'r comes from the slot of the
// current CG step (two slots, selected by F_STEP, which workgroup 0 of the A kernel bumps once per step).
// Body of one (virtual) block b of nb: shared with the persistent CG kernel (cg_persist.hpp).
// `alpha_of()` is called AFTER the first element of every lane's two loops has been requested: those loads do not depend on
// alpha, so the reduction of the partials that alpha comes from (a dependent chain of ~2 us) no longer runs in front of an idle
// memory pipe (round 4; same operations on the same operands in the same order: same bits).
template <class Sync, class AlphaFn>
__device__ __forceinline__ void cg_update_block(double *x, double *r, const double *__restrict__ p, const double *__restrict__ Gp,
                                                const double *__restrict__ M, int n, double *yacc, const double *__restrict__ z,
                                                int m, AlphaFn alpha_of, double *part, int b, int nb, int tid, double *sm, Sync sync,
                                                bool active = true, const double *__restrict__ Gp2 = nullptr) {
  double mx = 0., s = 0.;
  const long i0 = (long)b * kVecThreads + tid, stride = (long)nb * kVecThreads;
  const bool hy = active && yacc && i0 < m, hx = active && i0 < n;
  double y0 = 0., z0 = 0., x0 = 0., p0 = 0., r0 = 0., g0 = 0., m0 = 0.;
  if (hy) {
    y0 = yacc[i0];
    z0 = z[i0];
  }
  if (hx) {
    x0 = x[i0];
    p0 = p[i0];
    r0 = r[i0];
    g0 = Gp2 ? Gp[i0] + Gp2[i0] : Gp[i0];
    m0 = M[i0];
  }
  const double alpha = alpha_of();
  if (active) {
    if (hy) yacc[i0] = y0 + alpha * z0;
    if (yacc)
      for (long i = i0 + stride; i < m; i += stride) yacc[i] += alpha * z[i];
    if (hx) {
      x[i0] = x0 + alpha * p0;
      const double ri = r0 - alpha * g0;
      r[i0] = ri;
      mx = fmax(mx, abs_nan_inf(ri));
      s += (m0 * ri) * ri;
    }
    for (long i = i0 + stride; i < n; i += stride) {
      x[i] += alpha * p[i];
      const double ri = r[i] - alpha * (Gp2 ? Gp[i] + Gp2[i] : Gp[i]);
      r[i] = ri;
      mx = fmax(mx, abs_nan_inf(ri));
      s += (M[i] * ri) * ri;
    }
  }
  mx = group_max<kVecThreads>(mx, sm, tid, sync);
  s = group_sum<kVecThreads>(s, sm, tid, sync);
  if (active && tid == 0) {
    part[b] = mx;
    part[nb + b] = s;
  }
}
__device__ __forceinline__ void d_cg_update(double *x, double *r, const double *__restrict__ p,
                                            const double *__restrict__ Gp, const double *__restrict__ M, int n,
                                            double *yacc, const double *__restrict__ z, int m,
                                            const double *pgp_part, int pgp_np, double *sc, const int *fl,
                                            double *part, const double *__restrict__ Gp2) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc;
  auto alpha_of = [&]() {
    const double pGp = part_sum(pgp_part, pgp_np, sm);
    if (threadIdx.x == 0) {
      bc = sc[(fl[F_STEP] & 1) ? S_ZTR_B : S_ZTR] / pGp;
      if (blockIdx.x == 0) sc[S_ALPHA] = bc;
    }
    __syncthreads();
    return bc;
  };
  cg_update_block(x, r, p, Gp, M, n, yacc, z, m, alpha_of, part, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, sm, BlockSync{}, true, Gp2);
}
__global__ __launch_bounds__(kVecThreads) void k_cg_update(double *x, double *r, const double *__restrict__ p,
                                                           const double *__restrict__ Gp,
                                                           const double *__restrict__ M, int n, double *yacc,
                                                           const double *__restrict__ z, int m,
                                                           const double *pgp_part, int pgp_np, double *sc,
                                                           const int *fl, double *part,
                                                           const double *__restrict__ Gp2 = nullptr) {
  d_cg_update(x, r, p, Gp, M, n, yacc, z, m, pgp_part, pgp_np, sc, fl, part, Gp2);
}

// p = M r + beta p.  beta = z'r(new) / z'r(old) and the convergence test are formed in the prologue from the
// partials of k_cg_update (every workgroup reduces them in the same fixed order); workgroup 0 then does the
// bookkeeping for the step: new z'r into the OTHER slot (nobody reads that one during this launch),
// ||r||_inf, the CG step counter and the done flag (p is dead once the flag is set).
__device__ __forceinline__ void d_cg_dir(double *p, const double *__restrict__ r, const double *__restrict__ M,
                                         int n, const double *upd_part, int upd_np, double *sc, int *fl) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc[2];
  const int slot = fl[F_STEP] & 1;
  // the first element of every lane is requested before the partials are reduced (it does not depend on beta)
  const long i0 = (long)blockIdx.x * kVecThreads + threadIdx.x, stride = (long)gridDim.x * kVecThreads;
  const bool h0 = i0 < n;
  double m0 = 0., r0 = 0., p0 = 0.;
  if (h0) {
    m0 = M[i0];
    r0 = r[i0];
    p0 = p[i0];
  }
  {
    const double rn = part_max(upd_part, upd_np, sm);
    const double ztr = part_sum(upd_part + upd_np, upd_np, sm);
    if (threadIdx.x == 0) {
      bc[0] = ztr / sc[slot ? S_ZTR_B : S_ZTR];
      bc[1] = ztr;
      if (blockIdx.x == 0) {
        sc[S_RNORM] = rn;
        sc[S_BETA] = bc[0];
        sc[slot ? S_ZTR : S_ZTR_B] = ztr;
        fl[F_ITERS] += 1;
        if (rn < sc[S_TOL]) fl[F_DONE] = 1;
      }
    }
    __syncthreads();
  }
  const double beta = bc[0];
  if (h0) p[i0] = m0 * r0 + beta * p0;
  for (long i = i0 + stride; i < n; i += stride) p[i] = M[i] * r[i] + beta * p[i];
}
__global__ __launch_bounds__(kVecThreads) void k_cg_dir(double *p, const double *__restrict__ r,
                                                        const double *__restrict__ M, int n,
                                                        const double *upd_part, int upd_np, double *sc,
                                                        int *fl) {
  d_cg_dir(p, r, M, n, upd_part, upd_np, sc, fl);
}

// Small systems (n <= kCgFuseMaxN): k_cg_update and k_cg_dir as ONE launch.  A CG step of a 10 000-column system is four
// launch-bound kernels; the direction update needs beta, i.e. the z'r partials of ALL workgroups of the update — so every
// workgroup publishes its part (agent-scope release: the XCD L2s are not coherent with each other) and takes a ticket, and the
// LAST arriver does what k_cg_dir does, for the whole vector (n is small: 128 elements per lane at most).  Nothing waits.
// Same partials, same reduction order, same elementwise arithmetic as the two kernels: the same bits (the grouped path keeps two).
constexpr int kCgFuseMaxN = 32768;
__global__ __launch_bounds__(kVecThreads) void k_cg_update_dir(double *x, double *r, double *p, const double *__restrict__ Gp,
                                                               const double *__restrict__ M, int n, double *yacc,
                                                               const double *__restrict__ z, int m, const double *pgp_part, int pgp_np,
                                                               double *sc, int *fl, double *part, const double *__restrict__ Gp2,
                                                               unsigned *ticket) {
  if (fl[F_DONE]) return;
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc[2];
  __shared__ unsigned tk;
  const int nb = (int)gridDim.x;
  auto alpha_of = [&]() {
    const double pGp = part_sum(pgp_part, pgp_np, sm);
    if (threadIdx.x == 0) {
      bc[0] = sc[(fl[F_STEP] & 1) ? S_ZTR_B : S_ZTR] / pGp;
      if (blockIdx.x == 0) sc[S_ALPHA] = bc[0];
    }
    __syncthreads();
    return bc[0];
  };
  cg_update_block(x, r, p, Gp, M, n, yacc, z, m, alpha_of, part, (int)blockIdx.x, nb, (int)threadIdx.x, sm, BlockSync{}, true, Gp2);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (tk != (unsigned)nb - 1) return;
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int slot = fl[F_STEP] & 1;
  {
    const double rn = part_max(part, nb, sm);
    const double ztr = part_sum(part + nb, nb, sm);
    if (threadIdx.x == 0) {
      bc[1] = ztr / sc[slot ? S_ZTR_B : S_ZTR];
      sc[S_RNORM] = rn;
      sc[S_BETA] = bc[1];
      sc[slot ? S_ZTR : S_ZTR_B] = ztr;
      fl[F_ITERS] += 1;
      if (rn < sc[S_TOL]) fl[F_DONE] = 1;
    }
    __syncthreads();
  }
  const double beta = bc[1];
  for (int i = threadIdx.x; i < n; i += kVecThreads) p[i] = M[i] * r[i] + beta * p[i];
}

// R-weighted dots for the tau quadratic (root_plus): [p'Rg, p'Rp, p'Rmu, mu'Rg] over the first l-1 entries
__device__ __forceinline__ void d_tau_dots(const double *__restrict__ p, const double *__restrict__ mu,
                                           const double *__restrict__ g, const double *__restrict__ diag_r,
                                           long n, double *part, int *stall_fl) {
  // first kernel behind a CG chunk: in run-ahead mode (stall_fl = the flag array) an unconverged solve stalls the queue
  if (stall_fl) {
    if (stall_fl[F_STALL]) return;
    if (!stall_fl[F_DONE]) {
      __syncthreads();  // every lane of this workgroup has read the flags before they change
      if (threadIdx.x == 0) atomicExch(&stall_fl[F_STALL], 1);  // F_DONE is raised by k_stall_latch behind this kernel
      return;
    }
  }
  __shared__ double sm[kVecThreads / 64];
  double a = 0., b = 0., c = 0., d = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) {
    const double r = diag_r[i], pi = p[i], mi = mu[i], gi = g[i];
    a += pi * gi * r;
    b += pi * pi * r;
    c += pi * mi * r;
    d += mi * gi * r;
  }
  a = block_sum<kVecThreads>(a, sm);
  b = block_sum<kVecThreads>(b, sm);
  c = block_sum<kVecThreads>(c, sm);
  d = block_sum<kVecThreads>(d, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = a;
    part[gridDim.x + blockIdx.x] = b;
    part[2 * gridDim.x + blockIdx.x] = c;
    part[3 * gridDim.x + blockIdx.x] = d;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_tau_dots(const double *__restrict__ p,
                                                          const double *__restrict__ mu,
                                                          const double *__restrict__ g,
                                                          const double *__restrict__ diag_r, long n,
                                                          double *part, int *stall_fl) {
  d_tau_dots(p, mu, g, diag_r, n, part, stall_fl);
}

// g'Rg (cached per scale)
__device__ __forceinline__ void d_gg(const double *__restrict__ g, const double *__restrict__ diag_r, long n,
                                     double *part) {
  __shared__ double sm[kVecThreads / 64];
  double a = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads)
    a += g[i] * g[i] * diag_r[i];
  a = block_sum<kVecThreads>(a, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ __launch_bounds__(kVecThreads) void k_gg(const double *__restrict__ g,
                                                    const double *__restrict__ diag_r, long n, double *part) {
  d_gg(g, diag_r, n, part);
}
__device__ __forceinline__ void d_fin_store_sum(const double *part, int np, double *sc, int slot) {
  __shared__ double sm[kVecThreads / 64];
  const double s = part_sum(part, np, sm);
  if (threadIdx.x == 0) sc[slot] = s;
}
__global__ __launch_bounds__(kVecThreads) void k_fin_store_sum(const double *part, int np, double *sc,
                                                               int slot) {
  d_fin_store_sum(part, np, sc, slot);
}

// tau_tilde = positive root of the scalar quadratic from the partials of k_tau_dots
__device__ __forceinline__ double tau_root(double pg, double pp, double pmu, double mug, double eta, double tau_scale, double gg,
                                           int first_iter) {
  if (first_iter) return 1.0;
  const double a = tau_scale + gg;
  const double b = mug - 2 * pg - eta * tau_scale;
  const double c = pp - pmu;
  return (-b + sqrt(fmax(b * b - 4 * a * c, 0.))) / (2 * a);
}

// u_t -= tau_t g;  u = 2 u_t - v;  free / zero-cone / nonnegative rows finished here
//   x rows: identity.  zero-cone rows: dual cone is free -> identity.  l rows: max(.,0).
// tau_t is formed in the prologue: every workgroup reduces the four k_tau_dots partial arrays in the same
// fixed order (no single-workgroup finalize launch in between); workgroup 0 publishes it in sc[S_TAUT].
__device__ __forceinline__ void d_cone_pre(double *ut, double *u, const double *__restrict__ v,
                                           const double *__restrict__ g, int n, int m, int nz, int nl,
                                           const double *params, double *sc, const double *tau_part, int np,
                                           const double *__restrict__ diag_r, int *stall_fl) {
  if (stall_fl && stall_fl[F_STALL]) {  // queue stalled by k_tau_dots: also park the CG-step kernels queued behind
    if (blockIdx.x == 0 && threadIdx.x == 0) stall_fl[F_DONE] = 1;
    return;
  }
  const long l = (long)n + m + 1;
  const int first_iter = params[P_FIRST] != 0.;
  __shared__ double sm[kVecThreads / 64];
  __shared__ double bc;
  {
    const double pg = part_sum(tau_part, np, sm);
    const double pp = part_sum(tau_part + np, np, sm);
    const double pmu = part_sum(tau_part + 2 * np, np, sm);
    const double mug = part_sum(tau_part + 3 * np, np, sm);
    if (threadIdx.x == 0) {
      bc = tau_root(pg, pp, pmu, mug, v[l - 1], diag_r[l - 1], sc[S_GG], first_iter);
      if (blockIdx.x == 0) sc[S_TAUT] = bc;
    }
    __syncthreads();
  }
  const double taut = bc;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < l; i += (long)gridDim.x * kVecThreads) {
    if (i < l - 1) {
      const double t = ut[i] - taut * g[i];
      ut[i] = t;
      double w = 2 * t - v[i];
      if (i >= n + nz && i < (long)n + nz + nl) w = fmax(w, 0.);
      u[i] = w;
    } else {
      ut[i] = taut;
      u[i] = first_iter ? 1.0 : fmax(2 * taut - v[i], 0.);
    }
  }
}
__global__ __launch_bounds__(kVecThreads) void k_cone_pre(double *ut, double *u, const double *__restrict__ v,
                                                          const double *__restrict__ g, int n, int m, int nz,
                                                          int nl, const double *params, double *sc,
                                                          const double *tau_part, int np,
                                                          const double *__restrict__ diag_r, int *stall_fl) {
  d_cone_pre(ut, u, v, g, n, m, nz, nl, params, sc, tau_part, np, diag_r, stall_fl);
}

// rsk = R (v + u - 2 u_t)
__device__ __forceinline__ void d_rsk(double *rsk, const double *__restrict__ v, const double *__restrict__ u,
                                      const double *__restrict__ ut, const double *__restrict__ diag_r, long l) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < l; i += (long)gridDim.x * kVecThreads)
    rsk[i] = (v[i] + u[i] - 2 * ut[i]) * diag_r[i];
}
__global__ __launch_bounds__(kVecThreads) void k_rsk(double *rsk, const double *__restrict__ v,
                                                     const double *__restrict__ u,
                                                     const double *__restrict__ ut,
                                                     const double *__restrict__ diag_r, long l) {
  d_rsk(rsk, v, u, ut, diag_r, l);
}

// v += alpha (u - u_t); vpart[block] = partial ||v_new||^2 — same partition and order as k_sumsq, so the next
// iteration's k_prep normalises with the same bits as if k_sumsq had run
__device__ __forceinline__ void d_v_update(double *v, const double *__restrict__ u,
                                           const double *__restrict__ ut, double alpha, long l, double *vpart,
                                           const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double sm[kVecThreads / 64];
  double s = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < l; i += (long)gridDim.x * kVecThreads) {
    const double vi = v[i] + alpha * (u[i] - ut[i]);
    v[i] = vi;
    s += vi * vi;
  }
  s = block_sum<kVecThreads>(s, sm);
  if (threadIdx.x == 0) vpart[blockIdx.x] = s;
}
__global__ __launch_bounds__(kVecThreads) void k_v_update(double *v, const double *__restrict__ u,
                                                          const double *__restrict__ ut, double alpha, long l,
                                                          double *vpart, const int *stall) {
  d_v_update(v, u, ut, alpha, l, vpart, stall);
}

// after a scale update: v = rsk / R+ + 2 u_t - u
__device__ __forceinline__ void d_v_rescale(double *v, const double *__restrict__ rsk,
                                            const double *__restrict__ u, const double *__restrict__ ut,
                                            const double *__restrict__ diag_r, long l) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < l; i += (long)gridDim.x * kVecThreads)
    v[i] = rsk[i] / diag_r[i] + 2 * ut[i] - u[i];
}
__global__ __launch_bounds__(kVecThreads) void k_v_rescale(double *v, const double *__restrict__ rsk,
                                                           const double *__restrict__ u,
                                                           const double *__restrict__ ut,
                                                           const double *__restrict__ diag_r, long l) {
  d_v_rescale(v, rsk, u, ut, diag_r, l);
}

// diag_r = [rho_x (n) | 1/(1000 scale) (z rows) | 1/scale (other rows) | 10]
__device__ __forceinline__ void d_set_diag_r(double *diag_r, int n, int m, int nz, double rho_x, double scale) {
  const long l = (long)n + m + 1;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < l; i += (long)gridDim.x * kVecThreads) {
    double r;
    if (i < n) r = rho_x;
    else if (i < (long)n + nz) r = 1.0 / (1000. * scale);
    else if (i < l - 1) r = 1.0 / scale;
    else r = 10.0;
    diag_r[i] = r;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_set_diag_r(double *diag_r, int n, int m, int nz, double rho_x,
                                                            double scale) {
  d_set_diag_r(diag_r, n, m, nz, rho_x, scale);
}

// g rhs: g = [c ; -b]
__device__ __forceinline__ void d_g_rhs(double *g, const double *__restrict__ h, int n, int m) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < (long)n + m; i += (long)gridDim.x * kVecThreads)
    g[i] = i < n ? h[i] : -h[i];
}
__global__ __launch_bounds__(kVecThreads) void k_g_rhs(double *g, const double *__restrict__ h, int n, int m) {
  d_g_rhs(g, h, n, m);
}

// generic KKT rhs prep for a standalone solve: tmp = rhs_y / r_y, (rhs_x stays)
__device__ __forceinline__ void d_kkt_prep(const double *__restrict__ rhs, const double *__restrict__ diag_r,
                                           double *tmp, int n, int m) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < m; i += (long)gridDim.x * kVecThreads)
    tmp[i] = rhs[n + i] / diag_r[n + i];
}
__global__ __launch_bounds__(kVecThreads) void k_kkt_prep(const double *__restrict__ rhs,
                                                          const double *__restrict__ diag_r, double *tmp,
                                                          int n, int m) {
  d_kkt_prep(rhs, diag_r, tmp, n, m);
}
// y = (A x - rhs_y) / r_y, with ax already in `ax`
__device__ __forceinline__ void d_kkt_y(double *rhs, const double *__restrict__ ax,
                                        const double *__restrict__ diag_r, int n, int m) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < m; i += (long)gridDim.x * kVecThreads)
    rhs[n + i] = (ax[i] - rhs[n + i]) / diag_r[n + i];
}
__global__ __launch_bounds__(kVecThreads) void k_kkt_y(double *rhs, const double *__restrict__ ax,
                                                       const double *__restrict__ diag_r, int n, int m) {
  d_kkt_y(rhs, ax, diag_r, n, m);
}

// Jacobi preconditioner: M_j = 1 / (R_x,j + P_jj + sum_i A_ij^2 / R_y,i)  over CSC(A) columns
__device__ __forceinline__ void d_precond(const int *__restrict__ colptr, const int *__restrict__ rowidx,
                                          const double *__restrict__ val, const double *__restrict__ diag_r,
                                          const double *__restrict__ Pdiag, double *M, int n) {
  for (long j = (long)blockIdx.x * kVecThreads + threadIdx.x; j < n; j += (long)gridDim.x * kVecThreads) {
    double d = diag_r[j] + (Pdiag ? Pdiag[j] : 0.0);
    for (int p = colptr[j]; p < colptr[j + 1]; ++p) d += val[p] * val[p] / diag_r[n + rowidx[p]];
    M[j] = 1.0 / d;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_precond(const int *__restrict__ colptr,
                                                         const int *__restrict__ rowidx,
                                                         const double *__restrict__ val,
                                                         const double *__restrict__ diag_r,
                                                         const double *__restrict__ Pdiag, double *M, int n) {
  d_precond(colptr, rowidx, val, diag_r, Pdiag, M, n);
}

// final solution in original units: x = E x_hat/(sigma tau) ...; mode selects solved / infeasible / unbounded scaling
__device__ __forceinline__ void d_unnormalize(const double *__restrict__ u, const double *__restrict__ rsk,
                                              const double *__restrict__ D, const double *__restrict__ E,
                                              double sigma, double fx, double fy, double fs, int n, int m,
                                              double *x, double *y, double *s) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < (long)n + m; i += (long)gridDim.x * kVecThreads) {
    if (i < n) {
      x[i] = u[i] * (E ? E[i] / sigma : 1.0) * fx;
    } else {
      const long k = i - n;
      y[k] = u[i] * (D ? D[k] / sigma : 1.0) * fy;
      s[k] = rsk[i] / (D ? D[k] * sigma : 1.0) * fs;
    }
  }
}
__global__ __launch_bounds__(kVecThreads) void k_unnormalize(const double *__restrict__ u,
                                                             const double *__restrict__ rsk,
                                                             const double *__restrict__ D,
                                                             const double *__restrict__ E, double sigma,
                                                             double fx, double fy, double fs, int n, int m,
                                                             double *x, double *y, double *s) {
  d_unnormalize(u, rsk, D, E, sigma, fx, fy, fs, n, m, x, y, s);
}

// partials of a'b (fixed-order two-stage sum: the consumer adds part[0 .. gridDim) in order)
__device__ __forceinline__ void d_dot_part(const double *__restrict__ a, const double *__restrict__ b, long n,
                                           double *part) {
  __shared__ double sm[kVecThreads / 64];
  double s = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) s += a[i] * b[i];
  s = block_sum<kVecThreads>(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(kVecThreads) void k_dot_part(const double *__restrict__ a,
                                                          const double *__restrict__ b, long n, double *part) {
  d_dot_part(a, b, n, part);
}

// in-place final scaling of the device copies of the solution (a NaN factor marks a vector the status leaves undefined)
__device__ __forceinline__ void d_scale3(double *x, double *y, double *s, int n, int m, double fx, double fy,
                                         double fs) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < (long)n + m; i += (long)gridDim.x * kVecThreads) {
    if (i < n) x[i] = (fx != fx) ? fx : x[i] * fx;
    else {
      const long k = i - n;
      y[k] = (fy != fy) ? fy : y[k] * fy;
      s[k] = (fs != fs) ? fs : s[k] * fs;
    }
  }
}
__global__ __launch_bounds__(kVecThreads) void k_scale3(double *x, double *y, double *s, int n, int m,
                                                        double fx, double fy, double fs) {
  d_scale3(x, y, s, n, m, fx, fy, fs);
}

// per-iteration CSV diagnostics: [||u-u_t||_2^2, ||v-v_prev||_2^2, ||u-u_t||_inf, ||v-v_prev||_inf]
__global__ __launch_bounds__(kVecThreads) void k_diff_norms(const double *__restrict__ u, const double *__restrict__ ut,
                                                            const double *__restrict__ v, const double *__restrict__ vp,
                                                            long l, double *part) {
  __shared__ double sm[kVecThreads / 64];
  double a = 0., b = 0., c = 0., d = 0.;
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < l; i += (long)gridDim.x * kVecThreads) {
    const double du = u[i] - ut[i], dv = v[i] - vp[i];
    a += du * du;
    b += dv * dv;
    c = fmax(c, abs_nan_inf(du));
    d = fmax(d, abs_nan_inf(dv));
  }
  a = block_sum<kVecThreads>(a, sm);
  b = block_sum<kVecThreads>(b, sm);
  c = block_max<kVecThreads>(c, sm);
  d = block_max<kVecThreads>(d, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = a;
    part[gridDim.x + blockIdx.x] = b;
    part[2 * gridDim.x + blockIdx.x] = c;
    part[3 * gridDim.x + blockIdx.x] = d;
  }
}

// collapse k groups of np partials into out[0..k): sums for the first `nsum` groups, max for the rest
__device__ __forceinline__ void d_fin_multi(const double *part, int np, int nsum, int nmax, double *out) {
  __shared__ double sm[kVecThreads / 64];
  for (int k = 0; k < nsum + nmax; ++k) {
    const double v = k < nsum ? part_sum(part + (size_t)k * np, np, sm) : part_max(part + (size_t)k * np, np, sm);
    if (threadIdx.x == 0) out[k] = v;
    __syncthreads();
  }
}
__global__ __launch_bounds__(kVecThreads) void k_fin_multi(const double *part, int np, int nsum, int nmax,
                                                           double *out) {
  d_fin_multi(part, np, nsum, nmax, out);
}

}  // namespace scship
