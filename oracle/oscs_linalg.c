/*
 * oscs_linalg.c — ORACLE (test infrastructure): dense vector kernels and CSC
 * mat-vecs.  Restates the roles of scs_source/src/linalg.c and
 * scs_source/linsys/scs_matrix.c (named at R:meson.build:191,199-202; sources
 * absent).  Matrix layout: CSC {m,n,x,i,p}, R:scs/scsobject.h:594-605.
 */
#include "oscs.h"

scs_float o_dot(const scs_float *x, const scs_float *y, scs_int n) {
  scs_float s = 0.;
  O_PAR_SUM(n, s)
  for (scs_int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}

scs_float o_norm_inf(const scs_float *x, scs_int n) {
  scs_float mx = 0.;
  O_PAR_MAX(n, mx)
  for (scs_int i = 0; i < n; ++i) {
    scs_float a = OABS(x[i]);
    if (a > mx) mx = a;
  }
  return mx;
}

scs_float o_norm_2(const scs_float *x, scs_int n) { return sqrt(o_dot(x, x, n)); }

void o_axpy(scs_float *y, const scs_float *x, scs_float a, scs_int n) {
  O_PAR_FOR(n)
  for (scs_int i = 0; i < n; ++i) y[i] += a * x[i];
}

void o_scale(scs_float *x, scs_float a, scs_int n) {
  O_PAR_FOR(n)
  for (scs_int i = 0; i < n; ++i) x[i] *= a;
}

/* y += A x : column-major scatter */
void o_accum_by_a(const ScsMatrix *A, const scs_float *x, scs_float *y) {
  for (scs_int j = 0; j < A->n; ++j) {
    scs_float xj = x[j];
    for (scs_int p = A->p[j]; p < A->p[j + 1]; ++p) y[A->i[p]] += A->x[p] * xj;
  }
}

/* y += A' x : column-major gather */
void o_accum_by_atrans(const ScsMatrix *A, const scs_float *x, scs_float *y) {
  O_PAR_FOR(A->n)
  for (scs_int j = 0; j < A->n; ++j) {
    scs_float acc = 0.;
    for (scs_int p = A->p[j]; p < A->p[j + 1]; ++p) acc += A->x[p] * x[A->i[p]];
    y[j] += acc;
  }
}

/* y += P x with P symmetric, only the upper triangle stored
 * (the wrapper guarantees this: R:scs/py/__init__.py:163-166). */
void o_accum_by_p(const ScsMatrix *P, const scs_float *x, scs_float *y) {
  for (scs_int j = 0; j < P->n; ++j) {
    for (scs_int p = P->p[j]; p < P->p[j + 1]; ++p) {
      scs_int i = P->i[p];
      if (i > j) continue; /* ignore any lower-tri entry */
      y[i] += P->x[p] * x[j];
      if (i != j) y[j] += P->x[p] * x[i];
    }
  }
}

/* Gaussian elimination with partial pivoting on a column-major n*n system.
 * Returns the number of pivots that were numerically non-zero. */
scs_int o_dense_solve(scs_float *M, scs_float *rhs, scs_int n) {
  scs_int rank = 0;
  for (scs_int k = 0; k < n; ++k) {
    scs_int piv = k;
    scs_float mx = OABS(M[k + n * k]);
    for (scs_int i = k + 1; i < n; ++i) {
      scs_float a = OABS(M[i + n * k]);
      if (a > mx) { mx = a; piv = i; }
    }
    if (!(mx > 0.) || !isfinite(mx)) return rank;
    rank++;
    if (piv != k) {
      for (scs_int j = 0; j < n; ++j) {
        scs_float t = M[k + n * j]; M[k + n * j] = M[piv + n * j]; M[piv + n * j] = t;
      }
      scs_float t = rhs[k]; rhs[k] = rhs[piv]; rhs[piv] = t;
    }
    for (scs_int i = k + 1; i < n; ++i) {
      scs_float f = M[i + n * k] / M[k + n * k];
      if (f == 0.) continue;
      for (scs_int j = k + 1; j < n; ++j) M[i + n * j] -= f * M[k + n * j];
      rhs[i] -= f * rhs[k];
    }
  }
  for (scs_int k = n - 1; k >= 0; --k) {
    scs_float s = rhs[k];
    for (scs_int j = k + 1; j < n; ++j) s -= M[k + n * j] * rhs[j];
    rhs[k] = s / M[k + n * k];
  }
  return rank;
}
