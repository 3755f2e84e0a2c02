/*
 * oscs_core.c — ORACLE (test infrastructure): the ADMM / Douglas-Rachford loop
 * on the homogeneous self-dual embedding, residuals, termination, adaptive
 * scale, warm start, update.
 *
 * Restates scs_source/src/scs.c (named at R:meson.build:195; absent).  Entry
 * points mirror what the reference glue calls: scs_init R:scs/scsobject.h:903,
 * scs_solve :986, scs_update :1217, scs_finish :1240,
 * scs_set_default_settings :520.  Algorithm: SURVEY.md App. A (A.2 iteration,
 * A.3 scaling, A.7 termination), O'Donoghue 2021.
 *
 * Iteration (state v, scaling R = diag(diag_r)):
 *   u_t = (R + Q)^{-1} R v        linear system + scalar tau from a quadratic
 *   u   = Pi_{R^n x K* x R+}(2 u_t - v)
 *   rsk = R (v + u - 2 u_t)
 *   v  += alpha (u - u_t)
 */
#include "oscs.h"
#include <time.h>

typedef struct {
  scs_int last_iter;
  scs_float xt_p_x, xt_p_x_tau, ctx, ctx_tau, bty, bty_tau, pobj, dobj, gap, tau, kap;
  scs_float res_pri, res_dual, res_infeas, res_unbdd_p, res_unbdd_a;
  scs_float *ax, *ax_s, *px, *aty, *ax_s_btau, *px_aty_ctau;
} OResiduals;

typedef struct {
  scs_int m, n, l;
  int indirect;
  ScsMatrix A, P;  /* owned copies (normalised in place) */
  int has_P;
  scs_float *b_orig, *c_orig, *b_norm, *c_norm;
  scs_float nm_b_orig, nm_c_orig;
  ScsSettings stgs;
  scs_float scale;
  OConeWork *cone;
  OScaling *scal;
  OLinSys *p;
  OAa *accel;
  scs_float *u, *u_t, *v, *v_prev, *rsk, *h, *g, *ls_ws, *diag_r;
  ScsSolution xys_norm, xys_orig;
  OResiduals r_norm, r_orig;
  scs_float sum_log_scale_factor, aa_norm, setup_time;
  scs_int n_log_scale_factor, last_scale_update_iter, scale_updates;
  scs_int rejected_accel_steps, accepted_accel_steps;
} OWork;

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void oscs_set_default_settings(ScsSettings *s) {
  /* defaults: R:README.md:98-104 (AA); R:test/test_warm_start_consistency.py:228-241
   * (scale 0.1, rho_x 1e-6, alpha 1.5); banner R:notebooks/scs_benchmarks.ipynb cell 2 */
  s->normalize = 1;
  s->scale = 0.1;
  s->adaptive_scale = 1;
  s->rho_x = 1e-6;
  s->max_iters = 100000;
  s->eps_abs = 1e-4;
  s->eps_rel = 1e-4;
  s->eps_infeas = 1e-7;
  s->alpha = 1.5;
  s->time_limit_secs = 0.;
  s->verbose = 1;
  s->warm_start = 0;
  s->acceleration_lookback = 10;
  s->acceleration_interval = 10;
  s->acceleration_type_1 = 1;
  s->acceleration_regularization = 1e-8;
  s->acceleration_relaxation = 1.0;
  s->write_data_filename = NULL;
  s->log_csv_filename = NULL;
}

static void alloc_res(OResiduals *r, scs_int m, scs_int n) {
  memset(r, 0, sizeof(*r));
  r->last_iter = -1;
  r->ax = o_vec_calloc(m);
  r->ax_s = o_vec_calloc(m);
  r->ax_s_btau = o_vec_calloc(m);
  r->px = o_vec_calloc(n);
  r->aty = o_vec_calloc(n);
  r->px_aty_ctau = o_vec_calloc(n);
}
static void free_res(OResiduals *r) {
  free(r->ax); free(r->ax_s); free(r->ax_s_btau); free(r->px); free(r->aty); free(r->px_aty_ctau);
}

static void copy_matrix(ScsMatrix *dst, const ScsMatrix *src) {
  scs_int nnz = src->p[src->n];
  dst->m = src->m; dst->n = src->n;
  dst->x = (scs_float *)malloc(OMAX(nnz, 1) * sizeof(scs_float));
  dst->i = (scs_int *)malloc(OMAX(nnz, 1) * sizeof(scs_int));
  dst->p = (scs_int *)malloc((src->n + 1) * sizeof(scs_int));
  memcpy(dst->p, src->p, (src->n + 1) * sizeof(scs_int));
#ifdef OSCS_OMP
  /* by columns, partitioned as o_accum_by_atrans streams them (first touch, oscs.h) */
  O_PAR_FOR(src->n)
  for (scs_int j = 0; j < src->n; ++j)
    for (scs_int q = src->p[j]; q < src->p[j + 1]; ++q) { dst->x[q] = src->x[q]; dst->i[q] = src->i[q]; }
#else
  memcpy(dst->x, src->x, nnz * sizeof(scs_float));
  memcpy(dst->i, src->i, nnz * sizeof(scs_int));
#endif
}

static scs_int validate(const ScsData *d, const ScsCone *k, const ScsSettings *s) {
  scs_int j, p;
  if (d->m <= 0 || d->n <= 0) return -1;
  if (!d->A || !d->b || !d->c) return -1;
  const ScsMatrix *A = d->A;
  if (A->m != d->m || A->n != d->n) return -1;
  if (A->p[0] != 0) return -1;
  for (j = 0; j < A->n; ++j) {
    if (A->p[j + 1] < A->p[j]) return -1;
    for (p = A->p[j]; p < A->p[j + 1]; ++p)
      if (A->i[p] < 0 || A->i[p] >= A->m) return -1;
  }
  if (d->P) {
    const ScsMatrix *P = d->P;
    if (P->m != d->n || P->n != d->n) return -1;
    for (j = 0; j < P->n; ++j)
      for (p = P->p[j]; p < P->p[j + 1]; ++p)
        if (P->i[p] < 0 || P->i[p] >= P->n) return -1;
  }
  if (o_validate_cone(k) < 0) return -1;
  if (o_cone_dims(k) != d->m) return -1;
  if (s->max_iters <= 0 || s->eps_abs < 0 || s->eps_rel < 0 || s->eps_infeas < 0) return -1;
  if (s->alpha <= 0 || s->alpha >= 2 || s->rho_x <= 0 || s->scale <= 0) return -1;
  if (s->acceleration_interval <= 0 || s->acceleration_lookback < 0) return -1;
  return 0;
}

static void set_diag_r(OWork *w) {
  scs_int i;
  for (i = 0; i < w->n; ++i) w->diag_r[i] = w->stgs.rho_x;
  o_set_r_y(w->cone, w->scale, &w->diag_r[w->n]);
  w->diag_r[w->l - 1] = O_TAU_FACTOR;
}

/* g = (R + M)^{-1} h,  h = [c; b] */
static void update_work_cache(OWork *w) {
  memcpy(w->g, w->h, (w->l - 1) * sizeof(scs_float));
  o_scale(&w->g[w->n], -1., w->m);
  o_solve_lin_sys(w->p, w->g, NULL, O_CG_BEST_TOL);
}

void oscs_finish(void *work);

void *oscs_init(const ScsData *d, const ScsCone *k, const ScsSettings *stgs, int indirect) {
  double t0 = now_ms();
  if (!d || !k || !stgs || validate(d, k, stgs) < 0) return NULL;
  OWork *w = (OWork *)calloc(1, sizeof(OWork));
  scs_int m = d->m, n = d->n, l = n + m + 1;
  w->m = m; w->n = n; w->l = l; w->indirect = indirect;
  w->stgs = *stgs;
  w->stgs.write_data_filename = NULL;
  w->stgs.log_csv_filename = NULL;
  w->scale = stgs->scale;
  copy_matrix(&w->A, d->A);
  w->has_P = d->P != NULL;
  if (w->has_P) copy_matrix(&w->P, d->P);
  w->b_orig = (scs_float *)malloc(m * sizeof(scs_float));
  w->c_orig = (scs_float *)malloc(n * sizeof(scs_float));
  w->b_norm = (scs_float *)malloc(m * sizeof(scs_float));
  w->c_norm = (scs_float *)malloc(n * sizeof(scs_float));
  memcpy(w->b_orig, d->b, m * sizeof(scs_float));
  memcpy(w->c_orig, d->c, n * sizeof(scs_float));
  memcpy(w->b_norm, d->b, m * sizeof(scs_float));
  memcpy(w->c_norm, d->c, n * sizeof(scs_float));
  w->nm_b_orig = o_norm_inf(w->b_orig, m);
  w->nm_c_orig = o_norm_inf(w->c_orig, n);
  w->cone = o_init_cone(k, m);
  if (stgs->normalize) {
    w->scal = o_normalize_a_p(w->has_P ? &w->P : NULL, &w->A, w->cone);
    o_normalize_b_c(w->scal, w->b_norm, w->c_norm);
  }
  w->u = o_vec_calloc(l);
  w->u_t = o_vec_calloc(l);
  w->v = o_vec_calloc(l);
  w->v_prev = o_vec_calloc(l);
  w->rsk = o_vec_calloc(l);
  w->h = o_vec_calloc(l - 1);
  w->g = o_vec_calloc(l - 1);
  w->ls_ws = o_vec_calloc(l - 1);
  w->diag_r = o_vec_calloc(l);
  w->xys_norm.x = o_vec_calloc(n);
  w->xys_norm.y = o_vec_calloc(m);
  w->xys_norm.s = o_vec_calloc(m);
  w->xys_orig.x = o_vec_calloc(n);
  w->xys_orig.y = o_vec_calloc(m);
  w->xys_orig.s = o_vec_calloc(m);
  alloc_res(&w->r_norm, m, n);
  alloc_res(&w->r_orig, m, n);
  set_diag_r(w);
  w->p = o_init_lin_sys(&w->A, w->has_P ? &w->P : NULL, w->diag_r, indirect);
  if (!w->p) { oscs_finish(w); return NULL; }
  w->accel = o_aa_init(l, stgs->acceleration_lookback, stgs->acceleration_type_1,
                       stgs->acceleration_regularization, stgs->acceleration_relaxation,
                       O_AA_SAFEGUARD_FACTOR, O_AA_MAX_WEIGHT_NORM);
  memcpy(w->h, w->c_norm, n * sizeof(scs_float));
  memcpy(&w->h[n], w->b_norm, m * sizeof(scs_float));
  update_work_cache(w);
  w->setup_time = now_ms() - t0;
  return w;
}

void oscs_finish(void *work) {
  OWork *w = (OWork *)work;
  if (!w) return;
  free(w->A.x); free(w->A.i); free(w->A.p);
  if (w->has_P) { free(w->P.x); free(w->P.i); free(w->P.p); }
  free(w->b_orig); free(w->c_orig); free(w->b_norm); free(w->c_norm);
  o_free_cone(w->cone); o_free_scaling(w->scal); o_free_lin_sys(w->p); o_aa_free(w->accel);
  free(w->u); free(w->u_t); free(w->v); free(w->v_prev); free(w->rsk); free(w->h); free(w->g);
  free(w->ls_ws); free(w->diag_r);
  free(w->xys_norm.x); free(w->xys_norm.y); free(w->xys_norm.s);
  free(w->xys_orig.x); free(w->xys_orig.y); free(w->xys_orig.s);
  free_res(&w->r_norm); free_res(&w->r_orig);
  free(w);
}

scs_int oscs_update(void *work, const scs_float *b, const scs_float *c) {
  OWork *w = (OWork *)work;
  if (b) memcpy(w->b_orig, b, w->m * sizeof(scs_float));
  if (c) memcpy(w->c_orig, c, w->n * sizeof(scs_float));
  memcpy(w->b_norm, w->b_orig, w->m * sizeof(scs_float));
  memcpy(w->c_norm, w->c_orig, w->n * sizeof(scs_float));
  w->nm_b_orig = o_norm_inf(w->b_orig, w->m);
  w->nm_c_orig = o_norm_inf(w->c_orig, w->n);
  if (w->scal) o_normalize_b_c(w->scal, w->b_norm, w->c_norm);
  memcpy(w->h, w->c_norm, w->n * sizeof(scs_float));
  memcpy(&w->h[w->n], w->b_norm, w->m * sizeof(scs_float));
  update_work_cache(w);
  return 0;
}

static scs_float dot_r(const OWork *w, const scs_float *x, const scs_float *y) {
  scs_float ip = 0.;
  O_PAR_SUM(w->l, ip)
  for (scs_int i = 0; i < w->l - 1; ++i) ip += x[i] * y[i] * w->diag_r[i];
  return ip;
}

static scs_float root_plus(const OWork *w, const scs_float *p, const scs_float *mu, scs_float eta) {
  scs_float a, b, c, tau_scale = w->diag_r[w->l - 1];
  a = tau_scale + dot_r(w, w->g, w->g);
  b = dot_r(w, mu, w->g) - 2 * dot_r(w, p, w->g) - eta * tau_scale;
  c = dot_r(w, p, p) - dot_r(w, p, mu);
  return (-b + sqrt(OMAX(b * b - 4 * a * c, 0.))) / (2 * a);
}

static scs_int project_lin_sys(OWork *w, scs_int iter) {
  scs_int n = w->n, l = w->l, i, status;
  scs_float *warm = NULL, tol = -1.0;
  o_par_copy(w->u_t, w->v, l);
  O_PAR_FOR(l)
  for (i = 0; i < l - 1; ++i) w->u_t[i] *= (i < n ? 1 : -1) * w->diag_r[i];
  if (w->indirect) {
    scs_float nm_ws;
    warm = w->ls_ws;
    o_par_copy(warm, w->u, l - 1);
    o_axpy(warm, w->g, w->u[l - 1], l - 1);
    tol = OMIN(o_norm_inf(w->r_norm.ax_s_btau, w->m), o_norm_inf(w->r_norm.px_aty_ctau, w->n));
    nm_ws = o_norm_inf(warm, n) / pow((scs_float)iter + 1, O_CG_RATE);
    tol = O_CG_TOL_FACTOR * OMIN(tol, nm_ws);
    tol = OMAX(O_CG_BEST_TOL, tol);
  }
  status = o_solve_lin_sys(w->p, w->u_t, warm, tol);
  if (iter < O_FEASIBLE_ITERS) w->u_t[l - 1] = 1.;
  else w->u_t[l - 1] = root_plus(w, w->u_t, w->v, w->v[l - 1]);
  o_axpy(w->u_t, w->g, -w->u_t[l - 1], l - 1);
  return status;
}

static scs_int project_cones(OWork *w, scs_int iter) {
  scs_int i, n = w->n, l = w->l, status;
  O_PAR_FOR(l)
  for (i = 0; i < l; ++i) w->u[i] = 2 * w->u_t[i] - w->v[i];
  status = o_proj_dual_cone(&w->u[n], w->cone, &w->diag_r[n]);
  if (iter < O_FEASIBLE_ITERS) w->u[l - 1] = 1.0;
  else w->u[l - 1] = OMAX(w->u[l - 1], 0.);
  return status;
}

static void compute_residuals(OResiduals *r, scs_int m, scs_int n) {
  r->res_pri = SAFEDIV_POS(o_norm_inf(r->ax_s_btau, m), r->tau);
  r->res_dual = SAFEDIV_POS(o_norm_inf(r->px_aty_ctau, n), r->tau);
  r->res_unbdd_a = NAN;
  r->res_unbdd_p = NAN;
  r->res_infeas = NAN;
  if (r->ctx_tau < 0) {
    r->res_unbdd_a = SAFEDIV_POS(o_norm_inf(r->ax_s, m), -r->ctx_tau);
    r->res_unbdd_p = SAFEDIV_POS(o_norm_inf(r->px, n), -r->ctx_tau);
  }
  if (r->bty_tau < 0) r->res_infeas = SAFEDIV_POS(o_norm_inf(r->aty, n), -r->bty_tau);
}

static void unnormalize_residuals(OWork *w) {
  OResiduals *rn = &w->r_norm, *r = &w->r_orig;
  scs_int i, m = w->m, n = w->n;
  scs_float pd = w->scal->primal_scale * w->scal->dual_scale;
  r->last_iter = rn->last_iter;
  r->tau = rn->tau;
  r->kap = rn->kap / pd;
  r->bty_tau = rn->bty_tau / pd;
  r->ctx_tau = rn->ctx_tau / pd;
  r->xt_p_x_tau = rn->xt_p_x_tau / pd;
  r->xt_p_x = rn->xt_p_x / pd;
  r->ctx = rn->ctx / pd;
  r->bty = rn->bty / pd;
  r->pobj = rn->pobj / pd;
  r->dobj = rn->dobj / pd;
  r->gap = rn->gap / pd;
  for (i = 0; i < m; ++i) {
    scs_float f = 1. / (w->scal->D[i] * w->scal->primal_scale);
    r->ax[i] = rn->ax[i] * f;
    r->ax_s[i] = rn->ax_s[i] * f;
    r->ax_s_btau[i] = rn->ax_s_btau[i] * f;
  }
  for (i = 0; i < n; ++i) {
    scs_float f = 1. / (w->scal->E[i] * w->scal->dual_scale);
    r->aty[i] = rn->aty[i] * f;
    r->px[i] = rn->px[i] * f;
    r->px_aty_ctau[i] = rn->px_aty_ctau[i] * f;
  }
  compute_residuals(r, m, n);
}

static void populate_residual_struct(OWork *w, scs_int iter) {
  scs_int n = w->n, m = w->m, i;
  scs_float *x = w->xys_norm.x, *y = w->xys_norm.y, *s = w->xys_norm.s;
  OResiduals *r = &w->r_norm;
  if (r->last_iter == iter) return;
  r->last_iter = iter;
  memcpy(x, w->u, n * sizeof(scs_float));
  memcpy(y, &w->u[n], m * sizeof(scs_float));
  memcpy(s, &w->rsk[n], m * sizeof(scs_float));
  r->tau = OABS(w->u[n + m]);
  r->kap = OABS(w->rsk[n + m]);
  memset(r->ax, 0, m * sizeof(scs_float));
  o_accum_by_a(&w->A, x, r->ax);
  for (i = 0; i < m; ++i) {
    r->ax_s[i] = r->ax[i] + s[i];
    r->ax_s_btau[i] = r->ax_s[i] - w->b_norm[i] * r->tau;
  }
  memset(r->px, 0, n * sizeof(scs_float));
  if (w->has_P) {
    o_accum_by_p(&w->P, x, r->px);
    r->xt_p_x_tau = o_dot(r->px, x, n);
  } else {
    r->xt_p_x_tau = 0.;
  }
  memset(r->aty, 0, n * sizeof(scs_float));
  o_accum_by_atrans(&w->A, y, r->aty);
  for (i = 0; i < n; ++i) r->px_aty_ctau[i] = r->px[i] + r->aty[i] + w->c_norm[i] * r->tau;
  r->bty_tau = o_dot(y, w->b_norm, m);
  r->ctx_tau = o_dot(x, w->c_norm, n);
  r->bty = SAFEDIV_POS(r->bty_tau, r->tau);
  r->ctx = SAFEDIV_POS(r->ctx_tau, r->tau);
  r->xt_p_x = SAFEDIV_POS(r->xt_p_x_tau, r->tau * r->tau);
  r->gap = OABS(r->xt_p_x + r->ctx + r->bty);
  r->pobj = r->xt_p_x / 2. + r->ctx;
  r->dobj = -r->xt_p_x / 2. - r->bty;
  compute_residuals(r, m, n);
  memcpy(w->xys_orig.x, x, n * sizeof(scs_float));
  memcpy(w->xys_orig.y, y, m * sizeof(scs_float));
  memcpy(w->xys_orig.s, s, m * sizeof(scs_float));
  if (w->scal) {
    o_un_normalize_sol(w->scal, &w->xys_orig);
    unnormalize_residuals(w);
  } else {
    OResiduals *ro = &w->r_orig;
    scs_float *ax = ro->ax, *ax_s = ro->ax_s, *px = ro->px, *aty = ro->aty, *a3 = ro->ax_s_btau, *p3 = ro->px_aty_ctau;
    *ro = *r;
    ro->ax = ax; ro->ax_s = ax_s; ro->px = px; ro->aty = aty; ro->ax_s_btau = a3; ro->px_aty_ctau = p3;
    memcpy(ax, r->ax, m * sizeof(scs_float));
    memcpy(ax_s, r->ax_s, m * sizeof(scs_float));
    memcpy(a3, r->ax_s_btau, m * sizeof(scs_float));
    memcpy(px, r->px, n * sizeof(scs_float));
    memcpy(aty, r->aty, n * sizeof(scs_float));
    memcpy(p3, r->px_aty_ctau, n * sizeof(scs_float));
  }
}

static scs_int has_converged(OWork *w, scs_int iter) {
  OResiduals *r = &w->r_orig;
  scs_float eps_abs = w->stgs.eps_abs, eps_rel = w->stgs.eps_rel, eps_infeas = w->stgs.eps_infeas;
  scs_int m = w->m, n = w->n;
  if (r->tau > 0.) {
    scs_float grl = OMAX(OMAX(OABS(r->xt_p_x), OABS(r->ctx)), OABS(r->bty));
    scs_float prl = OMAX(OMAX(w->nm_b_orig * r->tau, o_norm_inf(w->xys_orig.s, m)), o_norm_inf(r->ax, m)) / r->tau;
    scs_float drl = OMAX(OMAX(w->nm_c_orig * r->tau, o_norm_inf(r->px, n)), o_norm_inf(r->aty, n)) / r->tau;
    if (isless(r->res_pri, eps_abs + eps_rel * prl) && isless(r->res_dual, eps_abs + eps_rel * drl) &&
        isless(r->gap, eps_abs + eps_rel * grl))
      return SCS_SOLVED;
  }
  if (isless(r->res_unbdd_a, eps_infeas) && isless(r->res_unbdd_p, eps_infeas) && iter > 0) return SCS_UNBOUNDED;
  if (isless(r->res_infeas, eps_infeas) && iter > 0) return SCS_INFEASIBLE;
  return 0;
}

static void update_scale(OWork *w, scs_int iter) {
  scs_int i, m = w->m, n = w->n;
  OResiduals *r = &w->r_orig;
  scs_int since = iter - w->last_scale_update_iter;
  scs_float factor, new_scale;
  scs_float nm_ax_s_btau = o_norm_inf(r->ax_s_btau, m), nm_px_aty_ctau = o_norm_inf(r->px_aty_ctau, n);
  scs_float rel_pri = SAFEDIV_POS(nm_ax_s_btau, OMAX(OMAX(o_norm_inf(r->ax, m), o_norm_inf(w->xys_orig.s, m)), w->nm_b_orig * r->tau));
  scs_float rel_dual = SAFEDIV_POS(nm_px_aty_ctau, OMAX(OMAX(o_norm_inf(r->px, n), o_norm_inf(r->aty, n)), w->nm_c_orig * r->tau));
  w->sum_log_scale_factor += log(rel_pri) - log(rel_dual);
  w->n_log_scale_factor++;
  factor = sqrt(exp(w->sum_log_scale_factor / (scs_float)(w->n_log_scale_factor)));
  if (since < O_RESCALING_MIN_ITERS) return;
  new_scale = OMIN(OMAX(w->scale * factor, O_MIN_SCALE_VALUE), O_MAX_SCALE_VALUE);
  if (new_scale == w->scale) return;
  if (factor > sqrt(10.) || factor < 1. / sqrt(10.)) {
    w->scale_updates++;
    w->sum_log_scale_factor = 0;
    w->n_log_scale_factor = 0;
    w->last_scale_update_iter = iter;
    w->scale = new_scale;
    set_diag_r(w);
    o_update_lin_sys_diag_r(w->p, w->diag_r);
    update_work_cache(w);
    if (w->accel) o_aa_reset(w->accel);
    /* keep rsk fixed under the new R:  R+ (v+ + u - 2u_t) = rsk */
    for (i = 0; i < n + m + 1; i++) w->v[i] = w->rsk[i] / w->diag_r[i] + 2 * w->u_t[i] - w->u[i];
  }
}

static void warm_start_vars(OWork *w, ScsSolution *sol) {
  scs_int n = w->n, m = w->m, i;
  scs_float *v = w->v;
  if (w->scal) o_normalize_sol(w->scal, sol);
  memcpy(v, sol->x, n * sizeof(scs_float));
  for (i = 0; i < m; ++i) v[i + n] = sol->y[i] + sol->s[i] / w->diag_r[i + n];
  v[n + m] = 1.0;
  for (i = 0; i < n + m + 1; ++i)
    if (!isfinite(v[i])) v[i] = 0.; /* a previous infeasible/unbounded solve leaves NaNs in sol */
  if (w->scal) o_un_normalize_sol(w->scal, sol);
}

static void set_solution(OWork *w, ScsSolution *sol, ScsInfo *info, scs_int iter) {
  scs_int n = w->n, m = w->m, i;
  OResiduals *r = &w->r_orig;
  memcpy(sol->x, w->u, n * sizeof(scs_float));
  memcpy(sol->y, &w->u[n], m * sizeof(scs_float));
  memcpy(sol->s, &w->rsk[n], m * sizeof(scs_float));
  if (w->scal) o_un_normalize_sol(w->scal, sol);
  populate_residual_struct(w, iter);
  info->iter = iter;
  info->res_infeas = r->res_infeas;
  info->res_unbdd_a = r->res_unbdd_a;
  info->res_unbdd_p = r->res_unbdd_p;
  info->scale = w->scale;
  info->scale_updates = w->scale_updates;
  info->rejected_accel_steps = w->rejected_accel_steps;
  info->accepted_accel_steps = w->accepted_accel_steps;
  info->comp_slack = OABS(o_dot(sol->s, sol->y, m));
  if (info->status_val == SCS_UNFINISHED) { /* hit max_iters / time limit: best guess */
    if (r->tau > r->kap) info->status_val = SCS_SOLVED_INACCURATE;
    else if (r->bty_tau < r->ctx_tau) info->status_val = SCS_INFEASIBLE_INACCURATE;
    else info->status_val = SCS_UNBOUNDED_INACCURATE;
  }
  switch (info->status_val) {
  case SCS_SOLVED:
  case SCS_SOLVED_INACCURATE: {
    scs_float it = SAFEDIV_POS(1.0, r->tau);
    o_scale(sol->x, it, n); o_scale(sol->y, it, m); o_scale(sol->s, it, m);
    info->gap = r->gap; info->res_pri = r->res_pri; info->res_dual = r->res_dual;
    info->pobj = r->xt_p_x / 2. + r->ctx;
    info->dobj = -r->xt_p_x / 2. - r->bty;
    strcpy(info->status, info->status_val == SCS_SOLVED ? "solved" : "solved (inaccurate - reached max_iters)");
    break;
  }
  case SCS_INFEASIBLE:
  case SCS_INFEASIBLE_INACCURATE:
    o_scale(sol->y, -1. / r->bty_tau, m);
    for (i = 0; i < n; ++i) sol->x[i] = NAN;
    for (i = 0; i < m; ++i) sol->s[i] = NAN;
    info->gap = NAN; info->res_pri = NAN; info->res_dual = NAN;
    info->pobj = INFINITY; info->dobj = INFINITY;
    strcpy(info->status, info->status_val == SCS_INFEASIBLE ? "infeasible" : "infeasible (inaccurate - reached max_iters)");
    break;
  default:
    o_scale(sol->x, -1. / r->ctx_tau, n);
    o_scale(sol->s, -1. / r->ctx_tau, m);
    for (i = 0; i < m; ++i) sol->y[i] = NAN;
    info->gap = NAN; info->res_pri = NAN; info->res_dual = NAN;
    info->pobj = -INFINITY; info->dobj = -INFINITY;
    strcpy(info->status, info->status_val == SCS_UNBOUNDED ? "unbounded" : "unbounded (inaccurate - reached max_iters)");
    break;
  }
}

scs_int oscs_solve(void *work, ScsSolution *sol, ScsInfo *info, scs_int warm_start) {
  OWork *w = (OWork *)work;
  scs_int i, n = w->n, m = w->m, l = w->l;
  double t_start = now_ms(), t_lin = 0, t_cone = 0, t_acc = 0, t;
  long cg0 = o_lin_sys_cg_iters(w->p);
  memset(info, 0, sizeof(*info));
  info->setup_time = w->setup_time;
  if (w->indirect) strcpy(info->lin_sys_solver, "oracle-cpu-indirect-cg");
  else snprintf(info->lin_sys_solver, sizeof(info->lin_sys_solver), "oracle-cpu-direct-ldl nnz(L)=%ld", o_lin_sys_nnz_l(w->p));
  /* per-solve state */
  w->sum_log_scale_factor = 0; w->n_log_scale_factor = 0; w->last_scale_update_iter = 0;
  w->scale_updates = 0; w->rejected_accel_steps = 0; w->accepted_accel_steps = 0; w->aa_norm = 0;
  w->r_norm.last_iter = -1; w->r_orig.last_iter = -1;
  memset(w->r_norm.ax_s_btau, 0, m * sizeof(scs_float));
  memset(w->r_norm.px_aty_ctau, 0, n * sizeof(scs_float));
  if (w->accel) o_aa_reset(w->accel);
  if (warm_start) {
    warm_start_vars(w, sol);
  } else {
    memset(w->v, 0, l * sizeof(scs_float));
    w->v[l - 1] = 1.;
  }
  memset(w->u, 0, l * sizeof(scs_float));
  w->u[l - 1] = 1.; /* so the first CG warm start is tau*g = g */
  info->status_val = SCS_UNFINISHED;

  for (i = 0; i < w->stgs.max_iters; ++i) {
    if (w->stgs.acceleration_lookback > 0 && i > 0 && i % w->stgs.acceleration_interval == 0) {
      t = now_ms();
      w->aa_norm = o_aa_apply(w->v, w->v_prev, w->accel);
      t_acc += now_ms() - t;
    }
    if (i >= O_FEASIBLE_ITERS) {
      scs_float nv = o_norm_2(w->v, l);
      o_scale(w->v, sqrt((scs_float)l) / OMAX(nv, 1e-300), l);
    }
    o_par_copy(w->v_prev, w->v, l);
    t = now_ms();
    if (project_lin_sys(w, i) < 0) { info->status_val = SCS_FAILED; break; }
    t_lin += now_ms() - t;
    t = now_ms();
    if (project_cones(w, i) < 0) { info->status_val = SCS_FAILED; break; }
    t_cone += now_ms() - t;
    O_PAR_FOR(l)
    for (scs_int j = 0; j < l; ++j) w->rsk[j] = (w->v[j] + w->u[j] - 2 * w->u_t[j]) * w->diag_r[j];
    if (i % O_CONVERGED_INTERVAL == 0) {
      populate_residual_struct(w, i);
      if ((info->status_val = has_converged(w, i)) != 0) break;
      if (w->stgs.time_limit_secs > 0 && (now_ms() - t_start) > 1e3 * w->stgs.time_limit_secs) break;
    }
    if (w->stgs.adaptive_scale && i == w->r_orig.last_iter) update_scale(w, i);
    O_PAR_FOR(l)
    for (scs_int j = 0; j < l; ++j) w->v[j] += w->stgs.alpha * (w->u[j] - w->u_t[j]);
    if (w->stgs.acceleration_lookback > 0 && i > 0 && i % w->stgs.acceleration_interval == 0) {
      t = now_ms();
      if (o_aa_safeguard(w->v, w->v_prev, w->accel) < 0) w->rejected_accel_steps++;
      else w->accepted_accel_steps++;
      t_acc += now_ms() - t;
    }
  }
  if (info->status_val == SCS_FAILED) {
    strcpy(info->status, "failure");
    for (scs_int j = 0; j < n; ++j) sol->x[j] = NAN;
    for (scs_int j = 0; j < m; ++j) sol->y[j] = sol->s[j] = NAN;
    info->iter = i;
  } else {
    set_solution(w, sol, info, i);
  }
  info->lin_sys_time = t_lin; info->cone_time = t_cone; info->accel_time = t_acc;
  info->cg_iters = (scs_int)(o_lin_sys_cg_iters(w->p) - cg0);
  if (w->accel) o_aa_get_stats(w->accel, &info->aa_stats);
  info->solve_time = now_ms() - t_start;
  return info->status_val;
}

const char *oscs_version(void) { return "3.2.11-oracle"; }

/* ---- kernel-level entry points used by the parity tests ---- */
scs_int oscs_proj_cone(scs_float *x, const ScsCone *k, scs_int m, int dual) {
  if (o_validate_cone(k) < 0 || o_cone_dims(k) != m) return -1;
  OConeWork *c = o_init_cone(k, m);
  scs_int st = dual ? o_proj_dual_cone(x, c, NULL) : o_proj_cone(x, c, NULL);
  o_free_cone(c);
  return st;
}

/* normalise a copy of (A,P,b,c); returns D,E,sigma — for parity of row a7 */
scs_int oscs_normalize(ScsMatrix *A, ScsMatrix *P, scs_float *b, scs_float *c, const ScsCone *k,
                       scs_float *D, scs_float *E, scs_float *sigma, scs_float *bl, scs_float *bu) {
  OConeWork *cw = o_init_cone(k, A->m);
  OScaling *s = o_normalize_a_p(P, A, cw);
  o_normalize_b_c(s, b, c);
  memcpy(D, s->D, A->m * sizeof(scs_float));
  memcpy(E, s->E, A->n * sizeof(scs_float));
  *sigma = s->primal_scale;
  if (k->bsize > 1 && bl && bu) {
    memcpy(bl, cw->k.bl, (k->bsize - 1) * sizeof(scs_float));
    memcpy(bu, cw->k.bu, (k->bsize - 1) * sizeof(scs_float));
  }
  o_free_scaling(s);
  o_free_cone(cw);
  return 0;
}

/* one KKT solve [[R_x+P, A'],[A,-R_y]] z = rhs (in place) — parity of row a4 */
scs_int oscs_kkt_solve(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, scs_float *rhs,
                       int indirect, scs_float tol, scs_int *cg_iters) {
  OLinSys *p = o_init_lin_sys(A, P, diag_r, indirect);
  if (!p) return -1;
  o_solve_lin_sys(p, rhs, NULL, tol);
  if (cg_iters) *cg_iters = (scs_int)o_lin_sys_cg_iters(p);
  o_free_lin_sys(p);
  return 0;
}

void oscs_spmv(const ScsMatrix *A, const scs_float *x, scs_float *y, int trans) {
  if (trans) o_accum_by_atrans(A, x, y);
  else o_accum_by_a(A, x, y);
}

/* all-core timing build only: the thread count of the NEXT workspace (its pages are first touched by these threads, oscs.h) */
void oscs_set_num_threads(int n) {
#ifdef OSCS_OMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
