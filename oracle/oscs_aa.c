/*
 * oscs_aa.c — ORACLE (test infrastructure): Anderson acceleration.
 *
 * Restates scs_source/src/aa.c (named at R:meson.build:187; absent).  Knobs and
 * defaults: R:README.md:98-104 (lookback 10, interval 10, type-I, regularisation
 * 1e-8, relaxation 1.0); statistics: R:scs/scsobject.h:1096-1107.
 *
 * Type-I :  gamma = (S'Y + r I)^{-1} S'g      Type-II:  gamma = (Y'Y + r I)^{-1} Y'g
 * with r = regularization * ||M||_F, next iterate f - D gamma (D = S - Y),
 * optional relaxation; a safeguard step rejects the extrapolation when the
 * fixed-point residual grew.
 */
#include "oscs.h"

struct OAa {
  scs_int type1, mem, dim, iter, success;
  scs_float relaxation, regularization, safeguard_factor, max_weight_norm;
  scs_float *x, *f, *g, *g_prev, *y, *s, *d, *Y, *S, *D, *M, *work, *x_work;
  scs_float norm_g;
  scs_float last_gamma[64]; /* weights of the most recent solve (tests) */
  scs_int last_len;
  ScsAaStats st;
};

OAa *o_aa_init(scs_int dim, scs_int mem, scs_int type1, scs_float regularization, scs_float relaxation,
               scs_float safeguard_factor, scs_float max_weight_norm) {
  OAa *a = (OAa *)calloc(1, sizeof(OAa));
  a->type1 = type1; a->mem = mem; a->dim = dim;
  a->relaxation = relaxation; a->regularization = regularization;
  a->safeguard_factor = safeguard_factor; a->max_weight_norm = max_weight_norm;
  if (mem <= 0) return a;
  size_t d = (size_t)dim;
  a->x = (scs_float *)calloc(d, sizeof(scs_float));
  a->f = (scs_float *)calloc(d, sizeof(scs_float));
  a->g = (scs_float *)calloc(d, sizeof(scs_float));
  a->g_prev = (scs_float *)calloc(d, sizeof(scs_float));
  a->y = (scs_float *)calloc(d, sizeof(scs_float));
  a->s = (scs_float *)calloc(d, sizeof(scs_float));
  a->d = (scs_float *)calloc(d, sizeof(scs_float));
  a->Y = (scs_float *)calloc(d * mem, sizeof(scs_float));
  a->S = (scs_float *)calloc(d * mem, sizeof(scs_float));
  a->D = (scs_float *)calloc(d * mem, sizeof(scs_float));
  a->M = (scs_float *)calloc((size_t)mem * mem, sizeof(scs_float));
  a->work = (scs_float *)calloc(OMAX(d, (size_t)mem), sizeof(scs_float));
  a->x_work = (relaxation != 1.0) ? (scs_float *)calloc(d, sizeof(scs_float)) : NULL;
  return a;
}

void o_aa_reset(OAa *a) { a->iter = 0; }

void o_aa_free(OAa *a) {
  if (!a) return;
  free(a->x); free(a->f); free(a->g); free(a->g_prev); free(a->y); free(a->s); free(a->d);
  free(a->Y); free(a->S); free(a->D); free(a->M); free(a->work); free(a->x_work);
  free(a);
}

void o_aa_get_stats(const OAa *a, ScsAaStats *st) { *st = a->st; }

scs_int o_aa_last_gamma(const OAa *a, scs_float *gamma) {
  if (gamma) memcpy(gamma, a->last_gamma, (size_t)a->last_len * sizeof(scs_float));
  return a->last_len;
}

static void set_m(OAa *a, scs_int len) {
  scs_int i, j, dim = a->dim;
  const scs_float *L = a->type1 ? a->S : a->Y;
  scs_float nrm = 0., r;
  for (j = 0; j < len; ++j)
    for (i = 0; i < len; ++i) {
      scs_float v = o_dot(&L[(size_t)i * dim], &a->Y[(size_t)j * dim], dim);
      a->M[i + len * j] = v;
      nrm += v * v;
    }
  r = a->regularization * sqrt(nrm);
  a->st.last_regularization = r;
  if (a->regularization > 0)
    for (i = 0; i < len; ++i) a->M[i + len * i] += r;
}

static void update_accel_params(const scs_float *x, const scs_float *f, OAa *a, scs_int len) {
  scs_int i, dim = a->dim, idx = (a->iter - 1) % a->mem;
  for (i = 0; i < dim; ++i) {
    a->g[i] = x[i] - f[i];
    a->s[i] = x[i] - a->x[i];
    a->d[i] = f[i] - a->f[i];
    a->y[i] = a->g[i] - a->g_prev[i];
  }
  memcpy(&a->S[(size_t)idx * dim], a->s, dim * sizeof(scs_float));
  memcpy(&a->D[(size_t)idx * dim], a->d, dim * sizeof(scs_float));
  memcpy(&a->Y[(size_t)idx * dim], a->y, dim * sizeof(scs_float));
  memcpy(a->f, f, dim * sizeof(scs_float));
  memcpy(a->x, x, dim * sizeof(scs_float));
  if (a->x_work) memcpy(a->x_work, x, dim * sizeof(scs_float));
  a->norm_g = o_norm_2(a->g, dim);
  memcpy(a->g_prev, a->g, dim * sizeof(scs_float));
  set_m(a, len);
}

static scs_float solve(scs_float *f, OAa *a, scs_int len) {
  scs_int i, j, dim = a->dim, rank;
  const scs_float *L = a->type1 ? a->S : a->Y;
  scs_float aa_norm;
  for (j = 0; j < len; ++j) a->work[j] = o_dot(&L[(size_t)j * dim], a->g, dim);
  rank = o_dense_solve(a->M, a->work, len);
  a->st.last_rank = rank;
  a->last_len = OMIN(len, 64);
  memcpy(a->last_gamma, a->work, (size_t)a->last_len * sizeof(scs_float));
  if (rank == 0) { a->st.n_reject_rank0++; a->success = 0; o_aa_reset(a); return -1.; }
  if (rank < len) { a->st.n_reject_lapack++; a->success = 0; o_aa_reset(a); return -1.; }
  aa_norm = o_norm_2(a->work, len);
  a->st.last_aa_norm = aa_norm;
  if (!isfinite(aa_norm)) { a->st.n_reject_nonfinite++; a->success = 0; o_aa_reset(a); return -1.; }
  if (aa_norm >= a->max_weight_norm) { a->st.n_reject_weight_cap++; a->success = 0; o_aa_reset(a); return -aa_norm; }
  /* f -= D * work */
  for (j = 0; j < len; ++j) {
    const scs_float *Dj = &a->D[(size_t)j * dim];
    scs_float wj = a->work[j];
    for (i = 0; i < dim; ++i) f[i] -= wj * Dj[i];
  }
  if (a->relaxation != 1.0) {
    /* x_work = x - S * work;  f = relaxation * f + (1 - relaxation) * x_work */
    for (j = 0; j < len; ++j) {
      const scs_float *Sj = &a->S[(size_t)j * dim];
      scs_float wj = a->work[j];
      for (i = 0; i < dim; ++i) a->x_work[i] -= wj * Sj[i];
    }
    for (i = 0; i < dim; ++i) f[i] = a->relaxation * f[i] + (1. - a->relaxation) * a->x_work[i];
  }
  a->success = 1;
  a->st.n_accept++;
  return aa_norm;
}

/* f = current map output F(x), x = map input; on return f may hold the AA iterate */
scs_float o_aa_apply(scs_float *f, const scs_float *x, OAa *a) {
  scs_float aa_norm = 0;
  scs_int len;
  a->success = 0;
  if (a->mem <= 0) return aa_norm;
  a->st.iter++;
  len = OMIN(a->iter, a->mem);
  if (a->iter == 0) {
    scs_int i;
    memcpy(a->x, x, a->dim * sizeof(scs_float));
    memcpy(a->f, f, a->dim * sizeof(scs_float));
    for (i = 0; i < a->dim; ++i) a->g_prev[i] = x[i] - f[i];
    a->iter++;
    return aa_norm;
  }
  update_accel_params(x, f, a, len);
  if (a->iter >= a->mem) aa_norm = solve(f, a, len);
  a->iter++;
  return aa_norm;
}

scs_int o_aa_safeguard(scs_float *f_new, scs_float *x_new, OAa *a) {
  scs_int i;
  scs_float norm_diff = 0.;
  if (!a->success) return 0;
  a->success = 0;
  for (i = 0; i < a->dim; ++i) {
    scs_float dlt = x_new[i] - f_new[i];
    norm_diff += dlt * dlt;
  }
  norm_diff = sqrt(norm_diff);
  if (norm_diff > a->safeguard_factor * a->norm_g) {
    memcpy(f_new, a->f, a->dim * sizeof(scs_float));
    memcpy(x_new, a->x, a->dim * sizeof(scs_float));
    a->st.n_safeguard_reject++;
    o_aa_reset(a);
    return -1;
  }
  return 0;
}
