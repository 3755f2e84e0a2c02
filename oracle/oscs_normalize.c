/*
 * oscs_normalize.c — ORACLE (test infrastructure): data equilibration.
 *
 * Restates scs_source/src/normalize.c + the normalisation helpers of
 * scs_source/linsys/scs_matrix.c (named at R:meson.build:192,200; absent).
 * `normalize` defaults to true (R:scs/scsobject.h:797).  Algorithm per SURVEY
 * App. A.6: 25 Ruiz (inf-norm) passes + 1 l2 pass on [P A'; A 0], row scalings
 * made constant inside each non-separable cone, factors clamped to
 * [1e-4, 1e4]; then b,c scaled by D,E and one scalar sigma.
 *
 *   A_hat = D A E,  P_hat = E P E,  b_hat = sigma D b,  c_hat = sigma E c
 *   x = E x_hat / sigma,  y = D y_hat / sigma,  s = s_hat / (D sigma)
 */
#include "oscs.h"

static scs_float apply_limit(scs_float x) {
  x = x < O_MIN_NORMALIZATION_FACTOR ? 1.0 : x;
  x = x > O_MAX_NORMALIZATION_FACTOR ? O_MAX_NORMALIZATION_FACTOR : x;
  return x;
}

static void compute_mats(const ScsMatrix *P, const ScsMatrix *A, scs_float *Dt, scs_float *Et,
                         const OConeWork *cone, int l2) {
  scs_int i, j, p;
  memset(Dt, 0, A->m * sizeof(scs_float));
  memset(Et, 0, A->n * sizeof(scs_float));
  /* rows of A -> D, cols of A -> E */
  for (j = 0; j < A->n; ++j) {
    for (p = A->p[j]; p < A->p[j + 1]; ++p) {
      scs_float v = OABS(A->x[p]);
      i = A->i[p];
      if (l2) { Dt[i] += v * v; Et[j] += v * v; }
      else { Dt[i] = OMAX(Dt[i], v); Et[j] = OMAX(Et[j], v); }
    }
  }
  /* symmetric P contributes to E through both its row and its column */
  if (P) {
    for (j = 0; j < P->n; ++j) {
      for (p = P->p[j]; p < P->p[j + 1]; ++p) {
        scs_float v = OABS(P->x[p]);
        i = P->i[p];
        if (i > j) continue;
        if (l2) { Et[j] += v * v; if (i != j) Et[i] += v * v; }
        else { Et[j] = OMAX(Et[j], v); Et[i] = OMAX(Et[i], v); }
      }
    }
  }
  if (l2) {
    for (i = 0; i < A->m; ++i) Dt[i] = sqrt(Dt[i]);
    for (j = 0; j < A->n; ++j) Et[j] = sqrt(Et[j]);
  }
  o_enforce_cone_boundaries(cone, Dt, l2 ? 1 : 0);
  for (i = 0; i < A->m; ++i) Dt[i] = SAFEDIV_POS(1.0, sqrt(apply_limit(Dt[i])));
  for (j = 0; j < A->n; ++j) Et[j] = SAFEDIV_POS(1.0, sqrt(apply_limit(Et[j])));
}

static void rescale(ScsMatrix *P, ScsMatrix *A, const scs_float *Dt, const scs_float *Et, OScaling *scal) {
  scs_int i, j, p;
  for (j = 0; j < A->n; ++j)
    for (p = A->p[j]; p < A->p[j + 1]; ++p) A->x[p] *= Dt[A->i[p]] * Et[j];
  if (P)
    for (j = 0; j < P->n; ++j)
      for (p = P->p[j]; p < P->p[j + 1]; ++p) P->x[p] *= Et[P->i[p]] * Et[j];
  for (i = 0; i < A->m; ++i) scal->D[i] *= Dt[i];
  for (j = 0; j < A->n; ++j) scal->E[j] *= Et[j];
}

/* box bounds follow the row scaling: bl_j <- bl_j D_{j+1}/D_0 */
static void normalize_box_cone(ScsCone *k, const scs_float *D, scs_int bsize) {
  for (scs_int j = 0; j < bsize - 1; j++) {
    if (k->bu[j] >= 1e15) k->bu[j] = INFINITY;
    else k->bu[j] = D ? D[j + 1] * k->bu[j] / D[0] : k->bu[j];
    if (k->bl[j] <= -1e15) k->bl[j] = -INFINITY;
    else k->bl[j] = D ? D[j + 1] * k->bl[j] / D[0] : k->bl[j];
  }
}

OScaling *o_normalize_a_p(ScsMatrix *P, ScsMatrix *A, OConeWork *cone) {
  scs_int i;
  OScaling *scal = (OScaling *)calloc(1, sizeof(OScaling));
  scs_float *Dt = (scs_float *)calloc(A->m, sizeof(scs_float));
  scs_float *Et = (scs_float *)calloc(A->n, sizeof(scs_float));
  scal->m = A->m;
  scal->n = A->n;
  scal->D = (scs_float *)calloc(A->m, sizeof(scs_float));
  scal->E = (scs_float *)calloc(A->n, sizeof(scs_float));
  for (i = 0; i < A->m; ++i) scal->D[i] = 1.;
  for (i = 0; i < A->n; ++i) scal->E[i] = 1.;
  for (i = 0; i < O_NUM_RUIZ_PASSES; ++i) {
    compute_mats(P, A, Dt, Et, cone, 0);
    rescale(P, A, Dt, Et, scal);
  }
  for (i = 0; i < O_NUM_L2_PASSES; ++i) {
    compute_mats(P, A, Dt, Et, cone, 1);
    rescale(P, A, Dt, Et, scal);
  }
  if (cone->k.bsize > 1) normalize_box_cone(&cone->k, &scal->D[cone->k.z + cone->k.l], cone->k.bsize);
  scal->primal_scale = scal->dual_scale = 1.;
  free(Dt);
  free(Et);
  return scal;
}

void o_normalize_b_c(OScaling *scal, scs_float *b, scs_float *c) {
  scs_int i;
  scs_float sigma;
  for (i = 0; i < scal->n; ++i) c[i] *= scal->E[i];
  for (i = 0; i < scal->m; ++i) b[i] *= scal->D[i];
  sigma = OMAX(o_norm_inf(c, scal->n), o_norm_inf(b, scal->m));
  sigma = sigma < O_MIN_NORMALIZATION_FACTOR ? 1.0 : sigma;
  sigma = sigma > O_MAX_NORMALIZATION_FACTOR ? O_MAX_NORMALIZATION_FACTOR : sigma;
  sigma = SAFEDIV_POS(1.0, sigma);
  o_scale(c, sigma, scal->n);
  o_scale(b, sigma, scal->m);
  scal->primal_scale = sigma;
  scal->dual_scale = sigma;
}

void o_normalize_sol(const OScaling *scal, ScsSolution *sol) {
  scs_int i;
  for (i = 0; i < scal->n; ++i) sol->x[i] /= (scal->E[i] / scal->dual_scale);
  for (i = 0; i < scal->m; ++i) sol->y[i] /= (scal->D[i] / scal->primal_scale);
  for (i = 0; i < scal->m; ++i) sol->s[i] *= (scal->D[i] * scal->dual_scale);
}

void o_un_normalize_sol(const OScaling *scal, ScsSolution *sol) {
  scs_int i;
  for (i = 0; i < scal->n; ++i) sol->x[i] *= (scal->E[i] / scal->dual_scale);
  for (i = 0; i < scal->m; ++i) sol->y[i] *= (scal->D[i] / scal->primal_scale);
  for (i = 0; i < scal->m; ++i) sol->s[i] /= (scal->D[i] * scal->dual_scale);
}

void o_free_scaling(OScaling *s) {
  if (!s) return;
  free(s->D);
  free(s->E);
  free(s);
}
