// cones.hpp — K5..K9: per-iteration projection onto the dual cone K*.
//
// Plays the role of scs_source/src/cones.c and exp_cone.c (named at
// R:meson.build:188,190; absent).  Cone order and slice layouts follow the
// reference's executable spec R:test/gen_random_cone_prob.py:90-130 (z,l,q,s,ep,ed,p)
// with the box cone between l and q (R:scs/scsobject.h:710-724).
//
// ADMM needs u_y = Pi_{K*}(w).  By Moreau, Pi_{K*}(w) = w + Pi_K(-w); for the
// self-dual cones (l, q, s) that is just Pi_K(w).  R_y is constant on every
// non-zero-cone row (k_set_diag_r), so the R-norm projection equals the
// Euclidean one on each cone.
//
// Mapping to the machine (gfx950, wave = 64):
//   z / l rows      — fused into the elementwise kernel k_cone_pre (vec.hpp)
//   SOC             — one wavefront per cone (shuffle-tree norm), many cones per workgroup;
//                     cones longer than kSocBig get a whole workgroup each
//   exp / pow       — one lane per 3-vector cone, bounded Newton iterations
//   box             — one workgroup, Newton on t with fixed-order block reductions
//   PSD             — one workgroup per matrix: parallel-order cyclic Jacobi eigensolve,
//                     then V diag(lambda+) V' (see psd.hpp)
#pragma once
#include "common.hpp"

namespace scship {

constexpr int kConeThreads = 256;
constexpr int kSocBig = 4096;

// ------------------------------------------------------------------ SOC
// in-place Pi_SOC on slices x[off[c] .. off[c]+dim[c]); one lane group of G = 8/16/32/64 lanes per cone, 64/G cones
// per wave (soc_group(): the narrowest group that holds the tail of the longest small cone, so short cones fill the
// wave instead of leaving 56 lanes idle).  The butterfly over a group adds in the same association order as the
// 64-lane tree does when the lanes beyond G hold zeros: the result does not depend on G.
__host__ __device__ inline int soc_group(int max_small_q) {
  const int tail = max_small_q - 1;
  return tail > 32 ? 64 : tail > 16 ? 32 : tail > 8 ? 16 : 8;
}
__device__ __forceinline__ void d_proj_soc_wave(double *x, const int *__restrict__ off,
                                                const int *__restrict__ dim, int ncones, int G, const int *stall, int blk) {
  SCS_STALL_GUARD(stall);
  const int lane = threadIdx.x & 63, gl = lane & (G - 1);
  const int wave = blk * (kConeThreads / 64) + (threadIdx.x >> 6);
  const int c = wave * (64 / G) + lane / G;
  const bool live = c < ncones;
  const int q = live ? dim[c] : 0;
  double *v = x + (live ? off[c] : 0);
  const bool small = q > 1 && q <= kSocBig;
  double ss = 0.;
  if (small)
    for (int i = 1 + gl; i < q; i += G) ss += v[i] * v[i];
  for (int o = G >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, kWave);  // every lane of the wave takes part
  if (q == 1 && gl == 0) v[0] = fmax(v[0], 0.);
  if (!small) return;
  const double s = sqrt(ss), t = v[0];
  if (s <= t) return;  // inside
  if (s <= -t) {
    for (int i = gl; i < q; i += G) v[i] = 0.;
    return;
  }
  const double alpha = 0.5 * (s + t), f = alpha / s;
  for (int i = 1 + gl; i < q; i += G) v[i] *= f;
  if (gl == 0) v[0] = alpha;
}
__global__ __launch_bounds__(kConeThreads) void k_proj_soc_wave(double *x, const int *__restrict__ off,
                                                                const int *__restrict__ dim, int ncones, int G,
                                                                const int *stall) {
  d_proj_soc_wave(x, off, dim, ncones, G, stall, (int)blockIdx.x);
}
// workgroups of k_proj_soc_wave for ncones cones in groups of G lanes
inline int soc_wave_blocks(int ncones, int G) { return ceil_div(ceil_div(ncones, 64 / G), kConeThreads / 64); }

// one workgroup per big cone
__global__ __launch_bounds__(kConeThreads) void k_proj_soc_block(double *x, const int *__restrict__ off,
                                                                 const int *__restrict__ dim, const int *__restrict__ big, int nbig, const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double sm[kConeThreads / 64];
  __shared__ double bc;
  const int c = big[blockIdx.x];
  const int q = dim[c];
  double *v = x + off[c];
  double ss = 0.;
  for (int i = 1 + threadIdx.x; i < q; i += kConeThreads) ss += v[i] * v[i];
  ss = block_sum<kConeThreads>(ss, sm);
  if (threadIdx.x == 0) bc = ss;
  __syncthreads();
  const double s = sqrt(bc), t = v[0];
  __syncthreads();
  if (s <= t) return;
  if (s <= -t) {
    for (int i = threadIdx.x; i < q; i += kConeThreads) v[i] = 0.;
    return;
  }
  const double alpha = 0.5 * (s + t), f = alpha / s;
  for (int i = 1 + threadIdx.x; i < q; i += kConeThreads) v[i] *= f;
  if (threadIdx.x == 0) v[0] = alpha;
}

// ------------------------------------------------------------ power cone
// {(x,y,z): x^a y^(1-a) >= |z|}; Newton on r (R:test/gen_random_cone_prob.py:176-231)
__device__ __forceinline__ double pow_calc_x(double r, double xh, double rh, double a) {
  return fmax(0.5 * (xh + sqrt(xh * xh + 4 * a * (rh - r) * r)), 1e-12);
}
__device__ inline void proj_power_cone(double *v, double a) {
  const double TOL = 1e-9;
  const double xh = v[0], yh = v[1], rh = fabs(v[2]);
  double x = 0., y = 0., r;
  if (xh >= 0 && yh >= 0 && TOL + pow(xh, a) * pow(yh, 1 - a) >= rh) return;
  if (xh <= 0 && yh <= 0 && TOL + pow(-xh, a) * pow(-yh, 1 - a) >= rh * pow(a, a) * pow(1 - a, 1 - a)) {
    v[0] = v[1] = v[2] = 0.;
    return;
  }
  r = rh / 2;
  for (int i = 0; i < 20; ++i) {
    x = pow_calc_x(r, xh, rh, a);
    y = pow_calc_x(r, yh, rh, 1 - a);
    const double xa_y1a = pow(x, a) * pow(y, 1 - a);
    const double f = xa_y1a - r;
    if (fabs(f) < TOL) break;
    const double dxdr = a * (rh - 2 * r) / (2 * x - xh);
    const double dydr = (1 - a) * (rh - 2 * r) / (2 * y - yh);
    const double fp = xa_y1a * (a * dxdr / x + (1 - a) * dydr / y) - 1;
    r = fmin(fmax(r - f / fp, 0.), rh);
  }
  v[0] = x;
  v[1] = y;
  v[2] = (v[2] < 0) ? -r : r;
}

// lane-per-cone: u = Pi_{K*}(w).  a >= 0: K = pow(a): u = w + Pi_K(-w);  a < 0: K* = pow(|a|): u = Pi_{pow(|a|)}(w)
__device__ __forceinline__ void d_proj_pow_dual(double *x, const double *__restrict__ a, int ncones,
                                                const int *stall) {
  SCS_STALL_GUARD(stall);
  const int c = blockIdx.x * kConeThreads + threadIdx.x;
  if (c >= ncones) return;
  double *w = x + 3L * c;
  const double ac = a[c];
  if (ac >= 0) {
    double t[3] = {-w[0], -w[1], -w[2]};
    proj_power_cone(t, ac);
    w[0] += t[0]; w[1] += t[1]; w[2] += t[2];
  } else {
    double t[3] = {w[0], w[1], w[2]};
    proj_power_cone(t, -ac);
    w[0] = t[0]; w[1] = t[1]; w[2] = t[2];
  }
}
__global__ __launch_bounds__(kConeThreads) void k_proj_pow_dual(double *x, const double *__restrict__ a,
                                                                int ncones, const int *stall) {
  d_proj_pow_dual(x, a, ncones, stall);
}
// primal-cone variant (test entry point): a >= 0: Pi_{pow(a)}(w);  a < 0: w + Pi_{pow(|a|)}(-w)
__global__ __launch_bounds__(kConeThreads) void k_proj_pow_primal(double *x, const double *__restrict__ a, int ncones, const int *stall) {
  SCS_STALL_GUARD(stall);
  const int c = blockIdx.x * kConeThreads + threadIdx.x;
  if (c >= ncones) return;
  double *w = x + 3L * c;
  const double ac = a[c];
  if (ac >= 0) {
    double t[3] = {w[0], w[1], w[2]};
    proj_power_cone(t, ac);
    w[0] = t[0]; w[1] = t[1]; w[2] = t[2];
  } else {
    double t[3] = {-w[0], -w[1], -w[2]};
    proj_power_cone(t, -ac);
    w[0] += t[0]; w[1] += t[1]; w[2] += t[2];
  }
}

// -------------------------------------------------------------- exp cone
// K_exp = cl{(r,s,t): s exp(r/s) <= t, s > 0}.  Univariate root-finding
// formulation of Friberg (2021): primal and polar projections share one root rho.
namespace expc {
constexpr double kInf = 1e15;
__device__ __forceinline__ double clip(double x, double lo, double hi) { return fmax(lo, fmin(hi, x)); }
__device__ __forceinline__ void hfun(const double *v0, double rho, double *f, double *df) {
  const double t0 = v0[2], s0 = v0[1], r0 = v0[0];
  const double e = exp(rho), en = exp(-rho);
  *f = ((rho - 1) * r0 + s0) * e - (r0 - rho * s0) * en - (rho * (rho - 1) + 1) * t0;
  *df = (rho * r0 + s0) * e + (r0 - (rho - 1) * s0) * en - (2 * rho - 1) * t0;
}
__device__ inline double root_binary(const double *v0, double xl, double xu, double x) {
  double xp = x, f, df;
  for (int i = 0; i < 80; ++i) {
    hfun(v0, x, &f, &df);
    if (f < 0.0) xl = x; else xu = x;
    xp = 0.5 * (xl + xu);
    if (fabs(xp - x) <= 1e-12 * fmax(1., fabs(xp)) || xp == xl || xp == xu) break;
    x = xp;
  }
  return xp;
}
__device__ inline double root_newton(const double *v0, double xl, double xu, double x) {
  const double EPS = 1e-15, DFTOL = 1e-13, LODAMP = 0.05, HIDAMP = 0.95;
  double xp, f, df;
  int i;
  for (i = 0; i < 20; ++i) {
    hfun(v0, x, &f, &df);
    if (fabs(f) <= EPS) break;
    if (f < 0.0) xl = x; else xu = x;
    if (xu <= xl) { xu = 0.5 * (xu + xl); xl = xu; break; }
    if (!isfinite(f) || df < DFTOL) break;
    xp = x - f / df;
    if (fabs(xp - x) <= EPS * fmax(1., fabs(xp))) break;
    if (xp >= xu) x = fmin(LODAMP * x + HIDAMP * xu, xu);
    else if (xp <= xl) x = fmax(LODAMP * x + HIDAMP * xl, xl);
    else x = xp;
  }
  if (i < 20) return clip(x, xl, xu);
  return root_binary(v0, xl, xu, x);
}
__device__ __forceinline__ double dist3(const double *a, const double *b) {
  const double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}
__device__ inline double primal_heur(const double *v0, double *vp) {
  const double t0 = v0[2], s0 = v0[1], r0 = v0[0];
  vp[2] = fmax(t0, 0.); vp[1] = 0.; vp[0] = fmin(r0, 0.);
  double dist = dist3(v0, vp);
  if (s0 > 0.) {
    const double tp = fmax(t0, s0 * exp(r0 / s0)), nd = tp - t0;
    if (nd < dist) { vp[2] = tp; vp[1] = s0; vp[0] = r0; dist = nd; }
  }
  return dist;
}
__device__ inline double polar_heur(const double *v0, double *vd) {
  const double t0 = v0[2], s0 = v0[1], r0 = v0[0];
  vd[2] = fmin(t0, 0.); vd[1] = fmin(s0, 0.); vd[0] = 0.;
  double dist = dist3(v0, vd);
  if (r0 > 0.) {
    const double td = fmin(t0, -r0 * exp(s0 / r0 - 1)), nd = t0 - td;
    if (nd < dist) { vd[2] = td; vd[1] = s0; vd[0] = r0; dist = nd; }
  }
  return dist;
}
__device__ __forceinline__ double ppsi(const double *v0) {
  const double s0 = v0[1], r0 = v0[0];
  const double q = sqrt(r0 * r0 + s0 * s0 - r0 * s0);
  const double psi = (r0 > s0) ? (r0 - s0 + q) / r0 : -s0 / (r0 - s0 - q);
  return ((psi - 1) * r0 + s0) / (psi * (psi - 1) + 1);
}
__device__ __forceinline__ double pomega(double rho) {
  double val = exp(rho) / (rho * (rho - 1) + 1);
  if (rho < 2.0) val = fmin(val, exp(2.0) / 3);
  return val;
}
__device__ __forceinline__ double dpsi(const double *v0) {
  const double s0 = v0[1], r0 = v0[0];
  const double q = sqrt(r0 * r0 + s0 * s0 - r0 * s0);
  const double psi = (s0 > r0) ? (r0 - q) / s0 : (r0 - s0) / (r0 + q);
  return (r0 - psi * s0) / (psi * (psi - 1) + 1);
}
__device__ __forceinline__ double domega(double rho) {
  double val = -exp(-rho) / (rho * (rho - 1) + 1);
  if (rho > -1.0) val = fmax(val, -exp(1.0) / 3);
  return val;
}
__device__ inline void bracket(const double *v0, double pdist, double ddist, double *lo, double *up) {
  const double t0 = v0[2], s0 = v0[1], r0 = v0[0];
  double baselow = -kInf, baseupr = kInf, low = -kInf, upr = kInf;
  const double mns = fmin(s0, 0.), mnr = fmin(r0, 0.);
  const double Dp = sqrt(fmax(pdist * pdist - mns * mns, 0.));
  const double Dd = sqrt(fmax(ddist * ddist - mnr * mnr, 0.));
  double cur, fl, fu, df;
  if (t0 > 0) { cur = log(t0 / ppsi(v0)); low = fmax(low, cur); }
  else if (t0 < 0) { cur = -log(-t0 / dpsi(v0)); upr = fmin(upr, cur); }
  if (r0 > 0) {
    baselow = 1 - s0 / r0;
    low = fmax(low, baselow);
    const double tpu = fmax(1e-12, fmin(Dd, Dp + t0));
    cur = fmax(low, baselow + tpu / r0 / pomega(low));
    upr = fmin(upr, cur);
  }
  if (s0 > 0) {
    baseupr = r0 / s0;
    upr = fmin(upr, baseupr);
    const double tdl = -fmax(1e-12, fmin(Dp, Dd - t0));
    cur = fmin(upr, baseupr - tdl / s0 / domega(upr));
    low = fmax(low, cur);
  }
  low = clip(fmin(low, upr), baselow, baseupr);
  upr = clip(fmax(low, upr), baselow, baseupr);
  if (low != upr) {
    hfun(v0, low, &fl, &df);
    hfun(v0, upr, &fu, &df);
    if (fl * fu > 0) {
      if (fabs(fl) < fabs(fu)) upr = low; else low = upr;
    }
  }
  *lo = low;
  *up = upr;
}
__device__ inline double sol_primal(const double *v0, double rho, double *vp) {
  const double lin = (rho - 1) * v0[0] + v0[1], e = exp(rho);
  if (lin > 0 && isfinite(e)) {
    const double q = rho * (rho - 1) + 1;
    vp[2] = e * lin / q; vp[1] = lin / q; vp[0] = rho * lin / q;
    return dist3(vp, v0);
  }
  vp[2] = kInf; vp[1] = 0.; vp[0] = 0.;
  return kInf;
}
__device__ inline double sol_polar(const double *v0, double rho, double *vd) {
  const double lin = v0[0] - rho * v0[1], e = exp(-rho);
  if (lin > 0 && isfinite(e)) {
    const double q = rho * (rho - 1) + 1, l = lin / q;
    vd[2] = -e * l; vd[1] = (1 - rho) * l; vd[0] = l;
    return dist3(v0, vd);
  }
  vd[2] = -kInf; vd[1] = 0.; vd[0] = 0.;
  return kInf;
}
// in-place projection onto K_exp (primal=1) or its dual (primal=0)
__device__ inline void proj(double *v0, int primal) {
  const double TOL = 1e-8;
  double xl, xh, vp[3], vd[3], vh[3];
  if (!primal) { v0[0] = -v0[0]; v0[1] = -v0[1]; v0[2] = -v0[2]; }
  double pdist = primal_heur(v0, vp), ddist = polar_heur(v0, vd);
  double err = fabs(vp[0] + vd[0] - v0[0]);
  err = fmax(err, fabs(vp[1] + vd[1] - v0[1]));
  err = fmax(err, fabs(vp[2] + vd[2] - v0[2]));
  bool opt = (v0[1] <= 0 && v0[0] <= 0);
  opt |= (fmin(pdist, ddist) <= TOL);
  opt |= (err <= TOL && (vp[0] * vd[0] + vp[1] * vd[1] + vp[2] * vd[2]) <= TOL);
  if (!opt) {
    bracket(v0, pdist, ddist, &xl, &xh);
    const double rho = root_newton(v0, xl, xh, 0.5 * (xl + xh));
    if (primal) {
      const double dh = sol_primal(v0, rho, vh);
      if (dh <= pdist) { vp[0] = vh[0]; vp[1] = vh[1]; vp[2] = vh[2]; }
    } else {
      const double dh = sol_polar(v0, rho, vh);
      if (dh <= ddist) { vd[0] = vh[0]; vd[1] = vh[1]; vd[2] = vh[2]; }
    }
  }
  if (primal) { v0[0] = vp[0]; v0[1] = vp[1]; v0[2] = vp[2]; }
  else { v0[0] = -vd[0]; v0[1] = -vd[1]; v0[2] = -vd[2]; }
}
}  // namespace expc

// lane-per-cone.  mode 0: u = Pi_{K*}(w) for the ep block (K = K_exp): dual projection;
//                 mode 1: ed block (K = K_exp^*): K* = K_exp: primal projection.
__device__ __forceinline__ void d_proj_exp(double *x, int ncones, int primal, const int *stall) {
  SCS_STALL_GUARD(stall);
  const int c = blockIdx.x * kConeThreads + threadIdx.x;
  if (c >= ncones) return;
  double *w = x + 3L * c;
  double t[3] = {w[0], w[1], w[2]};
  expc::proj(t, primal);
  w[0] = t[0]; w[1] = t[1]; w[2] = t[2];
}
__global__ __launch_bounds__(kConeThreads) void k_proj_exp(double *x, int ncones, int primal, const int *stall) {
  d_proj_exp(x, ncones, primal, stall);
}

// -------------------------------------------------------------- box cone
// K = {(t,s): t bl <= s <= t bu, t >= 0}.  In place: x <- x + Pi_K(-x) (dual=1) or Pi_K(x) (dual=0).
// One workgroup; Newton on t of a piecewise quadratic, warm-started from sc_t (device scalar).
constexpr int kBoxThreads = 1024;
__device__ __forceinline__ void d_proj_box(double *x, const double *__restrict__ bl,
                                           const double *__restrict__ bu, int bsize, double *t_warm, int dual,
                                           const int *stall) {
  SCS_STALL_GUARD(stall);
  __shared__ double sm[kBoxThreads / 64];
  __shared__ double bc[2];
  const double sgn = dual ? -1.0 : 1.0;
  if (bsize == 1) {
    if (threadIdx.x == 0) {
      const double w = sgn * x[0], p = fmax(w, 0.);
      x[0] = dual ? x[0] + p : p;
    }
    return;
  }
  const double t0 = sgn * x[0];
  double t = *t_warm;
  for (int iter = 0; iter < 25; ++iter) {
    double gt = 0., ht = 0.;
    for (int j = threadIdx.x; j < bsize - 1; j += kBoxThreads) {
      const double xj = sgn * x[1 + j], u = bu[j], l = bl[j];
      if (xj > t * u) { gt += (t * u - xj) * u; ht += u * u; }
      else if (xj < t * l) { gt += (t * l - xj) * l; ht += l * l; }
    }
    gt = block_sum<kBoxThreads>(gt, sm);
    ht = block_sum<kBoxThreads>(ht, sm);
    if (threadIdx.x == 0) {
      gt += (t - t0);
      ht += 1.0;
      const double tn = fmax(t - gt / fmax(ht, 1e-8), 0.);
      bc[0] = tn;
      bc[1] = (fabs(gt / (ht + 1e-6)) < 1e-9 || fabs(tn - t) < 1e-9) ? 1.0 : 0.0;
    }
    __syncthreads();
    t = bc[0];
    const bool stop = bc[1] != 0.0;
    __syncthreads();
    if (stop) break;
  }
  for (int j = threadIdx.x; j < bsize - 1; j += kBoxThreads) {
    const double xj = sgn * x[1 + j], u = bu[j], l = bl[j];
    double p = xj;
    if (xj > t * u) p = t * u;
    else if (xj < t * l) p = t * l;
    x[1 + j] = dual ? x[1 + j] + p : p;
  }
  if (threadIdx.x == 0) {
    x[0] = dual ? x[0] + t : t;
    *t_warm = t;
  }
}
__global__ __launch_bounds__(kBoxThreads) void k_proj_box(double *x, const double *__restrict__ bl,
                                                          const double *__restrict__ bu, int bsize,
                                                          double *t_warm, int dual, const int *stall) {
  d_proj_box(x, bl, bu, bsize, t_warm, dual, stall);
}

// Large box cones (bsize > kBoxMultiMin): the same Newton iteration on t, one launch per round over many workgroups.
// Every workgroup reduces its slice of the gradient / curvature sums, publishes the pair (write-through stores) and takes
// a ticket; the LAST arriver adds the pairs in slice order (fixed order: deterministic), takes the Newton step and
// leaves t and the stop flag for the next round's launch — no grid barrier, nothing spins.  state = {t, stop, t0-cache};
// the rounds after convergence return at once; k_proj_box_apply clips with the final t.  (One workgroup needs ~25 us per
// round for 1e5 bounds — a latency-bound 2.4 MB read — i.e. 283 us per projection; here a round is ~3 us.)
constexpr int kBoxMultiMin = 16384, kBoxMultiThreads = 256, kBoxMultiMaxWgs = 256, kBoxRounds = 25;
inline int box_multi_wgs(int bsize) { const int w = (bsize - 1 + 4 * kBoxMultiThreads - 1) / (4 * kBoxMultiThreads); return w < 1 ? 1 : (w > kBoxMultiMaxWgs ? kBoxMultiMaxWgs : w); }

__global__ __launch_bounds__(kBoxMultiThreads) void k_proj_box_round(const double *__restrict__ x, const double *__restrict__ bl,
                                                                     const double *__restrict__ bu, int bsize, double *state, double *parts,
                                                                     unsigned *ticket, int dual, int round, const int *stall) {
  SCS_STALL_GUARD(stall);
  if (round > 0 && state[1] != 0.) return;  // converged in an earlier round
  __shared__ double sm[kBoxMultiThreads / 64];
  __shared__ unsigned tk;
  const double sgn = dual ? -1.0 : 1.0;
  const double t = state[0];
  double gt = 0., ht = 0.;
  for (int j = blockIdx.x * kBoxMultiThreads + threadIdx.x; j < bsize - 1; j += gridDim.x * kBoxMultiThreads) {
    const double xj = sgn * x[1 + j], u = bu[j], l = bl[j];
    if (xj > t * u) { gt += (t * u - xj) * u; ht += u * u; }
    else if (xj < t * l) { gt += (t * l - xj) * l; ht += l * l; }
  }
  gt = block_sum<kBoxMultiThreads>(gt, sm);
  ht = block_sum<kBoxMultiThreads>(ht, sm);
  if (threadIdx.x == 0) {
    __hip_atomic_store(parts + 2 * blockIdx.x, gt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(parts + 2 * blockIdx.x + 1, ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (tk != gridDim.x - 1) return;  // (the last arriver leaves the ticket at 0 for the next round's launch: no wrap-around, ever)
  if (threadIdx.x == 0) {
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double g = 0., h = 0.;
    for (unsigned b = 0; b < gridDim.x; ++b) {
      g += __hip_atomic_load(parts + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      h += __hip_atomic_load(parts + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double t0 = sgn * x[0];
    g += (t - t0);
    h += 1.0;
    const double tn = fmax(t - g / fmax(h, 1e-8), 0.);
    state[0] = tn;
    state[1] = (fabs(g / (h + 1e-6)) < 1e-9 || fabs(tn - t) < 1e-9) ? 1.0 : 0.0;
  }
}

__global__ __launch_bounds__(kBoxMultiThreads) void k_proj_box_apply(double *x, const double *__restrict__ bl, const double *__restrict__ bu,
                                                                     int bsize, double *state, int dual, const int *stall) {
  SCS_STALL_GUARD(stall);
  const double sgn = dual ? -1.0 : 1.0;
  const double t = state[0];
  for (int j = blockIdx.x * kBoxMultiThreads + threadIdx.x; j < bsize - 1; j += gridDim.x * kBoxMultiThreads) {
    const double xj = sgn * x[1 + j], u = bu[j], l = bl[j];
    double p = xj;
    if (xj > t * u) p = t * u;
    else if (xj < t * l) p = t * l;
    x[1 + j] = dual ? x[1 + j] + p : p;
  }
  // x[0] is read by every workgroup of the LAST round launch and by nobody here: workgroup 0 may overwrite it
  if (blockIdx.x == 0 && threadIdx.x == 0) x[0] = dual ? x[0] + t : t;
}

// nonnegative / zero rows for the standalone projection entry point
__global__ __launch_bounds__(kConeThreads) void k_proj_zl(double *x, int nz, int nl, int dual) {
  const long i = (long)blockIdx.x * kConeThreads + threadIdx.x;
  if (i >= (long)nz + nl) return;
  if (i < nz) { if (!dual) x[i] = 0.; }
  else x[i] = fmax(x[i], 0.);
}

}  // namespace scship
