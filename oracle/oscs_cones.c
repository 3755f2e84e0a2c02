/*
 * oscs_cones.c — ORACLE (test infrastructure): cone projections.
 *
 * Restates scs_source/src/cones.c and exp_cone.c (named at R:meson.build:188,190;
 * sources absent).  Behavioural spec that IS in the reference:
 *   - cone order z,l,[box],q,s,[cs],ep,ed,p and slice layout: R:test/gen_random_cone_prob.py:90-130
 *   - SOC (t first):                                     R:test/gen_random_cone_prob.py:133-150
 *   - PSD vec = lower-tri col-major, off-diag * sqrt(2): R:test/gen_random_cone_prob.py:153-173
 *   - power cone Newton on r:                            R:test/gen_random_cone_prob.py:176-231
 *   - exp cone (r,s,t), s*exp(r/s) <= t:                 R:test/gen_random_cone_prob.py:234-315
 *   - box cone (t,s), bl*t <= s <= bu*t, dim len(bu)+1:  R:scs/scsobject.h:710-724,
 *                                                        R:test/test_scs_coverage.py:553-560
 *   - complex PSD cone `cs`, k*k reals per order-k Hermitian matrix, after `s`:
 *                                                        R:scs/scsobject.h:734-737,
 *                                                        R:test/test_spectral_and_complex_cones.py:22-24,
 *                                                        R:test/test_mix_sd_csd_cone.py:34-35
 *     (element order inside the k*k slice is NOT evidenced in the reference: UPSTREAM-RECALL of the
 *     SCS docs — lower triangle, column by column: H_jj, then sqrt2*Re H_ij, sqrt2*Im H_ij for i>j)
 * The exp-cone projection uses the univariate root-finding formulation of
 * Friberg (2021, "Projection onto the exponential cone: a univariate
 * root-finding problem"), which is what SCS >= 3.2 documents; the reference's
 * Python bisection is the golden it is checked against (tests/golden).
 */
#include "oscs.h"

#define CONE_TOL (1e-9)
#define BOX_CONE_MAX_ITERS (25)
#define POW_CONE_TOL (1e-9)
#define POW_CONE_MAX_ITERS (20)
#define MAX_BOX_VAL (1e15)
#define EXP_INF (1e15)

static scs_int sd_size(scs_int s) { return (s * (s + 1)) / 2; }

scs_int o_cone_dims(const ScsCone *k) {
  scs_int i, c = k->z + k->l + k->bsize;
  for (i = 0; i < k->qsize; ++i) c += k->q[i];
  for (i = 0; i < k->ssize; ++i) c += sd_size(k->s[i]);
  for (i = 0; i < k->cssize; ++i) c += k->cs[i] * k->cs[i];
  c += 3 * (k->ep + k->ed + k->psize);
  return c;
}

scs_int o_validate_cone(const ScsCone *k) {
  scs_int i;
  if (k->z < 0 || k->l < 0 || k->bsize < 0 || k->ep < 0 || k->ed < 0) return -1;
  for (i = 0; i < k->cssize; ++i)
    if (k->cs[i] < 0) return -1;
  for (i = 0; i < k->bsize - 1; ++i)
    if (k->bl[i] > k->bu[i]) return -1;
  for (i = 0; i < k->qsize; ++i)
    if (k->q[i] < 0) return -1;
  for (i = 0; i < k->ssize; ++i)
    if (k->s[i] < 0) return -1;
  for (i = 0; i < k->psize; ++i)
    if (k->p[i] < -1 || k->p[i] > 1) return -1;
  return 0;
}

OConeWork *o_init_cone(const ScsCone *k, scs_int m) {
  OConeWork *c = (OConeWork *)calloc(1, sizeof(OConeWork));
  scs_int i, cnt;
  c->k = *k;
  c->m = m;
  /* deep copies */
  if (k->bsize > 1) {
    c->k.bu = (scs_float *)malloc((k->bsize - 1) * sizeof(scs_float));
    c->k.bl = (scs_float *)malloc((k->bsize - 1) * sizeof(scs_float));
    memcpy(c->k.bu, k->bu, (k->bsize - 1) * sizeof(scs_float));
    memcpy(c->k.bl, k->bl, (k->bsize - 1) * sizeof(scs_float));
  } else {
    c->k.bu = c->k.bl = NULL;
  }
  c->k.q = (scs_int *)malloc(OMAX(k->qsize, 1) * sizeof(scs_int));
  if (k->qsize) memcpy(c->k.q, k->q, k->qsize * sizeof(scs_int));
  c->k.s = (scs_int *)malloc(OMAX(k->ssize, 1) * sizeof(scs_int));
  if (k->ssize) memcpy(c->k.s, k->s, k->ssize * sizeof(scs_int));
  c->k.p = (scs_float *)malloc(OMAX(k->psize, 1) * sizeof(scs_float));
  if (k->psize) memcpy(c->k.p, k->p, k->psize * sizeof(scs_float));
  c->k.cs = (scs_int *)malloc(OMAX(k->cssize, 1) * sizeof(scs_int));
  if (k->cssize) memcpy(c->k.cs, k->cs, k->cssize * sizeof(scs_int));
  /* boundaries: rows that can be scaled independently first, then one block per
   * non-separable cone */
  c->n_boundaries = 1 + k->qsize + k->ssize + k->cssize + k->ep + k->ed + k->psize;
  c->boundaries = (scs_int *)calloc(c->n_boundaries, sizeof(scs_int));
  cnt = 0;
  c->boundaries[cnt++] = k->z + k->l + k->bsize;
  for (i = 0; i < k->qsize; ++i) c->boundaries[cnt++] = k->q[i];
  for (i = 0; i < k->ssize; ++i) c->boundaries[cnt++] = sd_size(k->s[i]);
  for (i = 0; i < k->cssize; ++i) c->boundaries[cnt++] = k->cs[i] * k->cs[i];
  for (i = 0; i < k->ep + k->ed + k->psize; ++i) c->boundaries[cnt++] = 3;
  c->s = (scs_float *)calloc(OMAX(m, 1), sizeof(scs_float));
  c->box_t_warm = 1.;
  c->max_s = 0;
  for (i = 0; i < k->ssize; ++i) c->max_s = OMAX(c->max_s, k->s[i]);
  for (i = 0; i < k->cssize; ++i) c->max_s = OMAX(c->max_s, 2 * k->cs[i]); /* real embedding is 2k x 2k */
  if (c->max_s > 0) {
    size_t n2 = (size_t)c->max_s * c->max_s;
    c->Xs = (scs_float *)calloc(n2, sizeof(scs_float));
    c->Vs = (scs_float *)calloc(n2, sizeof(scs_float));
    c->es = (scs_float *)calloc(c->max_s, sizeof(scs_float));
  }
  return c;
}

void o_free_cone(OConeWork *c) {
  if (!c) return;
  free(c->k.bu); free(c->k.bl); free(c->k.q); free(c->k.s); free(c->k.cs); free(c->k.p);
  free(c->boundaries); free(c->s); free(c->Xs); free(c->Vs); free(c->es);
  free(c);
}

/* R_y: small weight on zero-cone rows (their dual is free), 1/scale elsewhere */
void o_set_r_y(const OConeWork *c, scs_float scale, scs_float *r_y) {
  scs_int i;
  for (i = 0; i < c->k.z; ++i) r_y[i] = 1.0 / (O_Z_CONE_R_FACTOR * scale);
  for (i = c->k.z; i < c->m; ++i) r_y[i] = 1.0 / scale;
}

/* make `vec` constant over each non-separable cone block (norm_inf or mean) */
void o_enforce_cone_boundaries(const OConeWork *c, scs_float *vec, int use_mean) {
  scs_int i, j, count = c->boundaries[0];
  for (i = 1; i < c->n_boundaries; ++i) {
    scs_int len = c->boundaries[i];
    if (len > 0) {
      scs_float w = 0.;
      if (use_mean) {
        for (j = 0; j < len; ++j) w += vec[count + j];
        w /= (scs_float)len;
      } else {
        w = o_norm_inf(&vec[count], len);
      }
      for (j = 0; j < len; ++j) vec[count + j] = w;
    }
    count += len;
  }
}

/* ------------------------------------------------------------------ SOC */
void o_proj_soc(scs_float *x, scs_int q) {
  if (q == 0) return;
  if (q == 1) { x[0] = OMAX(x[0], 0.); return; }
  scs_float v1 = x[0], s = o_norm_2(&x[1], q - 1), alpha;
  if (s <= v1) return;
  if (s <= -v1) { memset(x, 0, q * sizeof(scs_float)); return; }
  alpha = (s + v1) / 2.0;
  x[0] = alpha;
  o_scale(&x[1], alpha / s, q - 1);
}

/* ------------------------------------------------------------------ PSD */
void o_sym_eig(scs_float *A, scs_int n, scs_float *V, scs_float *e) {
  scs_int p, q, k, sweep;
  for (p = 0; p < n; ++p)
    for (q = 0; q < n; ++q) V[p + n * q] = (p == q) ? 1. : 0.;
  for (sweep = 0; sweep < 100; ++sweep) {
    scs_float off = 0., diag = 0.;
    for (p = 0; p < n; ++p) {
      diag += A[p + n * p] * A[p + n * p];
      for (q = p + 1; q < n; ++q) off += A[p + n * q] * A[p + n * q];
    }
    if (off <= 1e-32 * (diag + off) || off == 0.) break;
    for (p = 0; p < n - 1; ++p) {
      for (q = p + 1; q < n; ++q) {
        scs_float apq = A[p + n * q];
        if (OABS(apq) < 1e-300) continue;
        scs_float theta = (A[q + n * q] - A[p + n * p]) / (2. * apq);
        scs_float t = ((theta >= 0) ? 1. : -1.) / (OABS(theta) + sqrt(theta * theta + 1.));
        scs_float c = 1. / sqrt(t * t + 1.), s = t * c;
        for (k = 0; k < n; ++k) { /* columns p,q */
          scs_float akp = A[k + n * p], akq = A[k + n * q];
          A[k + n * p] = c * akp - s * akq;
          A[k + n * q] = s * akp + c * akq;
        }
        for (k = 0; k < n; ++k) { /* rows p,q */
          scs_float apk = A[p + n * k], aqk = A[q + n * k];
          A[p + n * k] = c * apk - s * aqk;
          A[q + n * k] = s * apk + c * aqk;
        }
        for (k = 0; k < n; ++k) {
          scs_float vkp = V[k + n * p], vkq = V[k + n * q];
          V[k + n * p] = c * vkp - s * vkq;
          V[k + n * q] = s * vkp + c * vkq;
        }
      }
    }
  }
  for (p = 0; p < n; ++p) e[p] = A[p + n * p];
}

scs_int o_proj_psd(scs_float *X, scs_int n, OConeWork *c) {
  scs_int i, j, k;
  const scs_float sqrt2 = sqrt(2.0), isqrt2 = 1.0 / sqrt(2.0);
  if (n == 0) return 0;
  if (n == 1) { X[0] = OMAX(X[0], 0.); return 0; }
  scs_float *Xs = c->Xs, *V = c->Vs, *e = c->es;
  /* unpack lower triangle (col-major), off-diagonals / sqrt(2) */
  k = 0;
  for (j = 0; j < n; ++j) {
    for (i = j; i < n; ++i) {
      scs_float v = X[k++];
      if (i != j) v *= isqrt2;
      Xs[i + n * j] = v;
      Xs[j + n * i] = v;
    }
  }
  o_sym_eig(Xs, n, V, e);
  /* X+ = sum_{e>0} e v v' */
  memset(Xs, 0, (size_t)n * n * sizeof(scs_float));
  for (k = 0; k < n; ++k) {
    if (e[k] <= 0) continue;
    const scs_float *v = &V[(size_t)n * k];
    for (j = 0; j < n; ++j) {
      scs_float ev = e[k] * v[j];
      for (i = j; i < n; ++i) Xs[i + n * j] += ev * v[i];
    }
  }
  k = 0;
  for (j = 0; j < n; ++j)
    for (i = j; i < n; ++i) X[k++] = (i == j) ? Xs[i + n * j] : Xs[i + n * j] * sqrt2;
  return 0;
}

/* Hermitian PSD cone.  H = A + iB (A symmetric, B antisymmetric) is PSD iff the real symmetric
 * M = [[A, -B], [B, A]] is, and Pi(M) = [[A+, -B+], [B+, A+]]: project the 2n x 2n embedding with the
 * real routine and read H+ back from its first block column. */
scs_int o_proj_cpsd(scs_float *X, scs_int n, OConeWork *c) {
  scs_int i, j, k, N = 2 * n;
  const scs_float sqrt2 = sqrt(2.0), isqrt2 = 1.0 / sqrt(2.0);
  if (n == 0) return 0;
  if (n == 1) { X[0] = OMAX(X[0], 0.); return 0; }
  scs_float *M = c->Xs, *V = c->Vs, *e = c->es;
  memset(M, 0, (size_t)N * N * sizeof(scs_float));
  k = 0;
  for (j = 0; j < n; ++j) {
    M[j + N * j] = M[(n + j) + N * (n + j)] = X[k++];
    for (i = j + 1; i < n; ++i) {
      const scs_float re = X[k++] * isqrt2, im = X[k++] * isqrt2;
      M[i + N * j] = M[j + N * i] = re;                         /* A */
      M[(n + i) + N * (n + j)] = M[(n + j) + N * (n + i)] = re; /* A */
      M[(n + i) + N * j] = M[j + N * (n + i)] = im;             /* B_ij  (lower-left block) */
      M[(n + j) + N * i] = M[i + N * (n + j)] = -im;            /* B_ji = -B_ij */
    }
  }
  o_sym_eig(M, N, V, e);
  k = 0;
  for (j = 0; j < n; ++j) {
    for (i = j; i < n; ++i) {
      scs_float re = 0., im = 0.;
      scs_int q;
      for (q = 0; q < N; ++q) {
        if (e[q] <= 0) continue;
        const scs_float *v = &V[(size_t)N * q];
        re += e[q] * v[i] * v[j];
        im += e[q] * v[n + i] * v[j];
      }
      if (i == j) X[k++] = re;
      else { X[k++] = re * sqrt2; X[k++] = im * sqrt2; }
    }
  }
  return 0;
}

/* ------------------------------------------------------------ power cone */
static scs_float pow_calc_x(scs_float r, scs_float xh, scs_float rh, scs_float a) {
  scs_float x = 0.5 * (xh + sqrt(xh * xh + 4 * a * (rh - r) * r));
  return OMAX(x, 1e-12);
}
static scs_float pow_calcdxdr(scs_float x, scs_float xh, scs_float rh, scs_float r, scs_float a) {
  return a * (rh - 2 * r) / (2 * x - xh);
}
static scs_float pow_calc_f(scs_float x, scs_float y, scs_float r, scs_float a) {
  return pow(x, a) * pow(y, (1 - a)) - r;
}
static scs_float pow_calc_fp(scs_float x, scs_float y, scs_float dxdr, scs_float dydr, scs_float a) {
  return pow(x, a) * pow(y, (1 - a)) * (a * dxdr / x + (1 - a) * dydr / y) - 1;
}

/* projection onto {(x,y,z): x^a y^(1-a) >= |z|, x,y >= 0}; R:test/gen_random_cone_prob.py:176-215 */
void o_proj_power_cone(scs_float *v, scs_float a) {
  scs_float xh = v[0], yh = v[1], rh = OABS(v[2]);
  scs_float x = 0.0, y = 0.0, r;
  scs_int i;
  /* v in K_a */
  if (xh >= 0 && yh >= 0 && POW_CONE_TOL + pow(xh, a) * pow(yh, (1 - a)) >= rh) return;
  /* -v in K_a^* */
  if (xh <= 0 && yh <= 0 &&
      POW_CONE_TOL + pow(-xh, a) * pow(-yh, 1 - a) >= rh * pow(a, a) * pow(1 - a, 1 - a)) {
    v[0] = v[1] = v[2] = 0;
    return;
  }
  r = rh / 2;
  for (i = 0; i < POW_CONE_MAX_ITERS; ++i) {
    scs_float f, fp, dxdr, dydr;
    x = pow_calc_x(r, xh, rh, a);
    y = pow_calc_x(r, yh, rh, 1 - a);
    f = pow_calc_f(x, y, r, a);
    if (OABS(f) < POW_CONE_TOL) break;
    dxdr = pow_calcdxdr(x, xh, rh, r, a);
    dydr = pow_calcdxdr(y, yh, rh, r, (1 - a));
    fp = pow_calc_fp(x, y, dxdr, dydr, a);
    r = OMAX(r - f / fp, 0);
    r = OMIN(r, rh);
  }
  v[0] = x;
  v[1] = y;
  v[2] = (v[2] < 0) ? -(r) : (r);
}

/* -------------------------------------------------------------- exp cone */
/* K_exp = cl{(r,s,t): s exp(r/s) <= t, s > 0}.  Friberg's formulation: project
 * onto K_exp and its polar simultaneously through one scalar root rho. */
static scs_float clipf(scs_float x, scs_float lo, scs_float hi) { return OMAX(lo, OMIN(hi, x)); }

static void hfun(const scs_float *v0, scs_float rho, scs_float *f, scs_float *df) {
  scs_float t0 = v0[2], s0 = v0[1], r0 = v0[0];
  scs_float exprho = exp(rho), expnegrho = exp(-rho);
  *f = ((rho - 1) * r0 + s0) * exprho - (r0 - rho * s0) * expnegrho - (rho * (rho - 1) + 1) * t0;
  *df = (rho * r0 + s0) * exprho + (r0 - (rho - 1) * s0) * expnegrho - (2 * rho - 1) * t0;
}

static scs_float root_search_binary(const scs_float *v0, scs_float xl, scs_float xu, scs_float x) {
  const scs_float EPS = 1e-12;
  scs_float x_plus = x, f, df;
  for (int i = 0; i < 80; i++) {
    hfun(v0, x, &f, &df);
    if (f < 0.0) xl = x; else xu = x;
    x_plus = 0.5 * (xl + xu);
    if (OABS(x_plus - x) <= EPS * OMAX(1., OABS(x_plus)) || (x_plus == xl) || (x_plus == xu)) break;
    x = x_plus;
  }
  return x_plus;
}

static scs_float root_search_newton(const scs_float *v0, scs_float xl, scs_float xu, scs_float x) {
  const scs_float EPS = 1e-15, DFTOL = 1e-13, LODAMP = 0.05, HIDAMP = 0.95;
  const int MAXITER = 20;
  scs_float x_plus, f, df;
  int i;
  for (i = 0; i < MAXITER; i++) {
    hfun(v0, x, &f, &df);
    if (OABS(f) <= EPS) break;
    if (f < 0.0) xl = x; else xu = x;
    if (xu <= xl) { xu = 0.5 * (xu + xl); xl = xu; break; }
    if (!isfinite(f) || df < DFTOL) break;
    x_plus = x - f / df;
    if (OABS(x_plus - x) <= EPS * OMAX(1., OABS(x_plus))) break;
    if (x_plus >= xu) x = OMIN(LODAMP * x + HIDAMP * xu, xu);
    else if (x_plus <= xl) x = OMAX(LODAMP * x + HIDAMP * xl, xl);
    else x = x_plus;
  }
  if (i < MAXITER) return clipf(x, xl, xu);
  return root_search_binary(v0, xl, xu, x);
}

static scs_float dist3(const scs_float *a, const scs_float *b) {
  return sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
}

static scs_float primal_heuristic(const scs_float *v0, scs_float *vp) {
  scs_float t0 = v0[2], s0 = v0[1], r0 = v0[0], dist, tp, newdist;
  vp[2] = OMAX(t0, 0); vp[1] = 0.0; vp[0] = OMIN(r0, 0);
  dist = dist3(v0, vp);
  if (s0 > 0.0) {
    tp = OMAX(t0, s0 * exp(r0 / s0));
    newdist = tp - t0;
    if (newdist < dist) { vp[2] = tp; vp[1] = s0; vp[0] = r0; dist = newdist; }
  }
  return dist;
}

static scs_float polar_heuristic(const scs_float *v0, scs_float *vd) {
  scs_float t0 = v0[2], s0 = v0[1], r0 = v0[0], dist, td, newdist;
  vd[2] = OMIN(t0, 0); vd[1] = OMIN(s0, 0); vd[0] = 0.0;
  dist = dist3(v0, vd);
  if (r0 > 0.0) {
    td = OMIN(t0, -r0 * exp(s0 / r0 - 1));
    newdist = t0 - td;
    if (newdist < dist) { vd[2] = td; vd[1] = s0; vd[0] = r0; dist = newdist; }
  }
  return dist;
}

static scs_float ppsi(const scs_float *v0) {
  scs_float s0 = v0[1], r0 = v0[0], psi;
  if (r0 > s0) psi = (r0 - s0 + sqrt(r0 * r0 + s0 * s0 - r0 * s0)) / r0;
  else psi = -s0 / (r0 - s0 - sqrt(r0 * r0 + s0 * s0 - r0 * s0));
  return ((psi - 1) * r0 + s0) / (psi * (psi - 1) + 1);
}
static scs_float pomega(scs_float rho) {
  scs_float val = exp(rho) / (rho * (rho - 1) + 1);
  if (rho < 2.0) val = OMIN(val, exp(2.0) / 3);
  return val;
}
static scs_float dpsi(const scs_float *v0) {
  scs_float s0 = v0[1], r0 = v0[0], psi;
  if (s0 > r0) psi = (r0 - sqrt(r0 * r0 + s0 * s0 - r0 * s0)) / s0;
  else psi = (r0 - s0) / (r0 + sqrt(r0 * r0 + s0 * s0 - r0 * s0));
  return (r0 - psi * s0) / (psi * (psi - 1) + 1);
}
static scs_float domega(scs_float rho) {
  scs_float val = -exp(-rho) / (rho * (rho - 1) + 1);
  if (rho > -1.0) val = OMAX(val, -exp(1.0) / 3);
  return val;
}

static void exp_search_bracket(const scs_float *v0, scs_float pdist, scs_float ddist,
                               scs_float *low_out, scs_float *upr_out) {
  scs_float t0 = v0[2], s0 = v0[1], r0 = v0[0];
  scs_float baselow = -EXP_INF, baseupr = EXP_INF, low = -EXP_INF, upr = EXP_INF;
  scs_float mns = OMIN(s0, 0), mnr = OMIN(r0, 0);
  scs_float Dp = sqrt(OMAX(pdist * pdist - mns * mns, 0.));
  scs_float Dd = sqrt(OMAX(ddist * ddist - mnr * mnr, 0.));
  scs_float curbnd, fl, fu, df, tpu, tdl;
  if (t0 > 0) {
    curbnd = log(t0 / ppsi(v0));
    low = OMAX(low, curbnd);
  } else if (t0 < 0) {
    curbnd = -log(-t0 / dpsi(v0));
    upr = OMIN(upr, curbnd);
  }
  if (r0 > 0) {
    baselow = 1 - s0 / r0;
    low = OMAX(low, baselow);
    tpu = OMAX(1e-12, OMIN(Dd, Dp + t0));
    curbnd = OMAX(low, baselow + tpu / r0 / pomega(low));
    upr = OMIN(upr, curbnd);
  }
  if (s0 > 0) {
    baseupr = r0 / s0;
    upr = OMIN(upr, baseupr);
    tdl = -OMAX(1e-12, OMIN(Dp, Dd - t0));
    curbnd = OMIN(upr, baseupr - tdl / s0 / domega(upr));
    low = OMAX(low, curbnd);
  }
  low = clipf(OMIN(low, upr), baselow, baseupr);
  upr = clipf(OMAX(low, upr), baselow, baseupr);
  if (low != upr) {
    hfun(v0, low, &fl, &df);
    hfun(v0, upr, &fu, &df);
    if (fl * fu > 0) {
      if (OABS(fl) < OABS(fu)) upr = low; else low = upr;
    }
  }
  *low_out = low;
  *upr_out = upr;
}

static scs_float sol_primal(const scs_float *v0, scs_float rho, scs_float *vp) {
  scs_float linrho = (rho - 1) * v0[0] + v0[1], exprho = exp(rho), quadrho;
  if (linrho > 0 && isfinite(exprho)) {
    quadrho = rho * (rho - 1) + 1;
    vp[2] = exprho * linrho / quadrho;
    vp[1] = linrho / quadrho;
    vp[0] = rho * linrho / quadrho;
    return dist3(vp, v0);
  }
  vp[2] = EXP_INF; vp[1] = 0.0; vp[0] = 0.0;
  return EXP_INF;
}

static scs_float sol_polar(const scs_float *v0, scs_float rho, scs_float *vd) {
  scs_float linrho = v0[0] - rho * v0[1], exprho = exp(-rho), quadrho, l;
  if (linrho > 0 && isfinite(exprho)) {
    quadrho = rho * (rho - 1) + 1;
    l = linrho / quadrho;
    vd[2] = -exprho * l;
    vd[1] = (1 - rho) * l;
    vd[0] = l;
    return dist3(v0, vd);
  }
  vd[2] = -EXP_INF; vd[1] = 0.0; vd[0] = 0.0;
  return EXP_INF;
}

/* in-place projection onto K_exp (primal=1) or K_exp^* (primal=0) */
void o_proj_exp_cone(scs_float *v0, int primal) {
  const scs_float TOL = 1e-8;
  scs_float xl, xh, pdist, ddist, err, rho, dist_hat, vp[3], vd[3], v_hat[3];
  int opt;
  if (!primal) { v0[0] *= -1.; v0[1] *= -1.; v0[2] *= -1.; } /* Pi_{K*}(v) = -Pi_{K°}(-v) */
  pdist = primal_heuristic(v0, vp);
  ddist = polar_heuristic(v0, vd);
  err = OABS(vp[0] + vd[0] - v0[0]);
  err = OMAX(err, OABS(vp[1] + vd[1] - v0[1]));
  err = OMAX(err, OABS(vp[2] + vd[2] - v0[2]));
  opt = (v0[1] <= 0 && v0[0] <= 0);
  opt |= (OMIN(pdist, ddist) <= TOL);
  opt |= (err <= TOL && (vp[0] * vd[0] + vp[1] * vd[1] + vp[2] * vd[2]) <= TOL);
  if (!opt) {
    exp_search_bracket(v0, pdist, ddist, &xl, &xh);
    rho = root_search_newton(v0, xl, xh, 0.5 * (xl + xh));
    if (primal) {
      dist_hat = sol_primal(v0, rho, v_hat);
      if (dist_hat <= pdist) { memcpy(vp, v_hat, sizeof(vp)); pdist = dist_hat; }
    } else {
      dist_hat = sol_polar(v0, rho, v_hat);
      if (dist_hat <= ddist) { memcpy(vd, v_hat, sizeof(vd)); ddist = dist_hat; }
    }
  }
  if (primal) {
    memcpy(v0, vp, sizeof(vp));
  } else {
    v0[0] = -vd[0]; v0[1] = -vd[1]; v0[2] = -vd[2];
  }
}

/* -------------------------------------------------------------- box cone */
/* project (t,s) onto {t*bl <= s <= t*bu, t >= 0} under the metric diag(r_box);
 * 1-D Newton on t of a piecewise quadratic */
static scs_float proj_box_cone(scs_float *tx, const scs_float *bl, const scs_float *bu, scs_int bsize,
                               scs_float t_warm_start, const scs_float *r_box) {
  scs_float *x, gt, ht, t_prev, t = t_warm_start, rho_t = 1, r;
  const scs_float *rho = NULL;
  scs_int iter, j;
  if (bsize == 1) { tx[0] = OMAX(tx[0], 0.0); return tx[0]; }
  x = &(tx[1]);
  if (r_box) { rho_t = 1.0 / r_box[0]; rho = &(r_box[1]); }
  for (iter = 0; iter < BOX_CONE_MAX_ITERS; iter++) {
    t_prev = t;
    gt = rho_t * (t - tx[0]);
    ht = rho_t;
    for (j = 0; j < bsize - 1; j++) {
      r = rho ? 1.0 / rho[j] : 1.;
      if (x[j] > t * bu[j]) {
        gt += r * (t * bu[j] - x[j]) * bu[j];
        ht += r * bu[j] * bu[j];
      } else if (x[j] < t * bl[j]) {
        gt += r * (t * bl[j] - x[j]) * bl[j];
        ht += r * bl[j] * bl[j];
      }
    }
    t = OMAX(t - gt / OMAX(ht, 1e-8), 0.);
    if (OABS(gt / (ht + 1e-6)) < CONE_TOL || OABS(t - t_prev) < CONE_TOL) break;
  }
  for (j = 0; j < bsize - 1; j++) {
    if (x[j] > t * bu[j]) x[j] = t * bu[j];
    else if (x[j] < t * bl[j]) x[j] = t * bl[j];
  }
  tx[0] = t;
  return t;
}

/* ------------------------------------------------------------ full cone */
scs_int o_proj_cone(scs_float *x, OConeWork *c, const scs_float *r_y) {
  const ScsCone *k = &c->k;
  scs_int i, count = 0;
  if (k->z) { o_par_zero(x, k->z); count += k->z; }
  O_PAR_FOR(k->l)
  for (i = count; i < count + k->l; ++i) x[i] = OMAX(x[i], 0.0);
  count += k->l;
  if (k->bsize) {
    c->box_t_warm = proj_box_cone(&x[count], k->bl, k->bu, k->bsize, c->box_t_warm,
                                  r_y ? &r_y[count] : NULL);
    count += k->bsize;
  }
#ifdef OSCS_OMP
  if (k->qsize > 1024) {  /* all-core timing variant: independent cones in parallel (offsets by a prefix sum) */
    scs_int *qoff = (scs_int *)malloc(((size_t)k->qsize + 1) * sizeof(scs_int));
    qoff[0] = count;
    for (i = 0; i < k->qsize; ++i) qoff[i + 1] = qoff[i] + k->q[i];
    O_PAR_FOR(k->qsize)
    for (i = 0; i < k->qsize; ++i) o_proj_soc(&x[qoff[i]], k->q[i]);
    count = qoff[k->qsize];
    free(qoff);
  } else
#endif
  for (i = 0; i < k->qsize; ++i) { o_proj_soc(&x[count], k->q[i]); count += k->q[i]; }
  for (i = 0; i < k->ssize; ++i) {
    o_proj_psd(&x[count], k->s[i], c);
    count += sd_size(k->s[i]);
  }
  for (i = 0; i < k->cssize; ++i) {
    o_proj_cpsd(&x[count], k->cs[i], c);
    count += k->cs[i] * k->cs[i];
  }
  O_PAR_FOR(k->ep)
  for (i = 0; i < k->ep; ++i) o_proj_exp_cone(&x[count + 3 * i], 1);
  count += 3 * k->ep;
  O_PAR_FOR(k->ed)
  for (i = 0; i < k->ed; ++i) o_proj_exp_cone(&x[count + 3 * i], 0);
  count += 3 * k->ed;
  for (i = 0; i < k->psize; ++i) {
    scs_float *v = &x[count];
    if (k->p[i] >= 0) {
      o_proj_power_cone(v, k->p[i]);
    } else { /* dual power cone via Moreau; R:test/gen_random_cone_prob.py:122-129 */
      scs_float w[3] = {-v[0], -v[1], -v[2]};
      o_proj_power_cone(w, -k->p[i]);
      v[0] += w[0]; v[1] += w[1]; v[2] += w[2];
    }
    count += 3;
  }
  return 0;
}

/* Moreau under the R-norm:  Pi_{K*}(x) = x + R^{-1} Pi_K(-R x)   */
scs_int o_proj_dual_cone(scs_float *x, OConeWork *c, const scs_float *r_y) {
  scs_int i, st;
  o_par_copy(c->s, x, c->m);
  O_PAR_FOR(c->m)
  for (i = 0; i < c->m; ++i) x[i] *= r_y ? -r_y[i] : -1.;
  st = o_proj_cone(x, c, r_y);
  O_PAR_FOR(c->m)
  for (i = 0; i < c->m; ++i) x[i] = (r_y ? x[i] / r_y[i] : x[i]) + c->s[i];
  return st;
}
