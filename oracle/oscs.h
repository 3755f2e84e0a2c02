/*
 * oscs.h — internal header of the CPU ORACLE (test infrastructure only).
 *
 * ORACLE STATUS: this directory is a CPU restatement of the SCS 3.2.x ADMM hot
 * path (cvxgrp/scs, pulled by the reference as the git submodule `scs_source`,
 * R:.gitmodules:1-3, presumed tag 3.2.11 from R:pyproject.toml:10).  The
 * submodule is EMPTY in /root/reference, so no reference source or binary can
 * be compiled here (`oracle/_ref` is unbuildable).  The
 * restatement follows the published algorithm (O'Donoghue, "Operator splitting
 * for a homogeneous embedding of the linear complementarity problem", 2021;
 * SCS docs) and is pinned against
 *   (1) golden vectors captured from R:test/gen_random_cone_prob.py
 *       (tests/golden/ npz files, generator tests/golden/make_golden.py), and
 *   (2) the closed-form / certificate answers of the reference's own tests.
 * Iterate-level parity with the upstream C core: "parity unpinned".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in oracle/.  The product (scs-python_amd/) never does.
 */
#ifndef OSCS_H_GUARD
#define OSCS_H_GUARD

#include "../include/scs_types.h"
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* OSCS_OMP (oracle/Makefile target liboscs_omp.so): the all-core TIMING variant of the oracle (bench.py cpu_baseline).
 * Element-wise loops and the mat-vecs run on OpenMP threads (per-row / per-column sums keep their order), reductions
 * become tree sums: results agree with the sequential checker to rounding, not bit for bit — the tests use liboscs.so. */
#ifdef OSCS_OMP
#include <omp.h>
#define O_PRAGMA(x) _Pragma(#x)
#define O_PAR_FOR(n) O_PRAGMA(omp parallel for schedule(static) if ((n) > 16384))
#define O_PAR_SUM(n, v) O_PRAGMA(omp parallel for schedule(static) reduction(+ : v) if ((n) > 16384))
#define O_PAR_MAX(n, v) O_PRAGMA(omp parallel for schedule(static) reduction(max : v) if ((n) > 16384))
/* First touch (round 4): on a two-socket host a page lives on the NUMA node of the thread that first writes it.  The timing
 * build therefore zero-fills / copies its long vectors and its matrix copies in PARALLEL loops with the same static partition as
 * the loops that later stream them (o_vec_calloc, o_par_copy*, and the matrix copies in oscs_linsys.c / oscs_core.c) — until
 * round 3 everything was first touched by the master thread.  (What capped the all-core leg at ~32 threads on the bench boxes,
 * profiles/r03_cpu_threads.txt, turned out to be the container's CFS quota of 16 CPUs, not placement: bench.py cpu_quota.)
 * The sequential checker build maps these to calloc / memcpy / memset. */
#else
#define O_PAR_FOR(n)
#define O_PAR_SUM(n, v)
#define O_PAR_MAX(n, v)
#endif
#include <stdlib.h>
#include <string.h>
static inline void o_par_zero(scs_float *x, scs_int n) {
#ifdef OSCS_OMP
  O_PAR_FOR(n)
  for (scs_int i = 0; i < n; ++i) x[i] = 0.;
#else
  memset(x, 0, (size_t)n * sizeof(scs_float));
#endif
}
static inline void o_par_copy(scs_float *dst, const scs_float *src, scs_int n) {
#ifdef OSCS_OMP
  O_PAR_FOR(n)
  for (scs_int i = 0; i < n; ++i) dst[i] = src[i];
#else
  memcpy(dst, src, (size_t)n * sizeof(scs_float));
#endif
}
/* zero-initialised vector whose pages are first touched by the threads that will stream them */
static inline scs_float *o_vec_calloc(scs_int n) {
#ifdef OSCS_OMP
  scs_float *x = (scs_float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(scs_float));
  if (x) o_par_zero(x, n);
  return x;
#else
  return (scs_float *)calloc((size_t)(n > 0 ? n : 1), sizeof(scs_float));
#endif
}

#define OMAX(a, b) (((a) > (b)) ? (a) : (b))
#define OMIN(a, b) (((a) < (b)) ? (a) : (b))
#define OABS(x) (((x) < 0) ? -(x) : (x))
#define SAFEDIV_POS(X, Y) ((Y) < 1e-18 ? ((X) / 1e-18) : (X) / (Y))

/* constants of the restated algorithm (SCS docs) */
#define O_TAU_FACTOR (10.)
#define O_Z_CONE_R_FACTOR (1000.)
#define O_FEASIBLE_ITERS (1)
#define O_CONVERGED_INTERVAL (25)
#define O_RESCALING_MIN_ITERS (100)
#define O_MIN_SCALE_VALUE (1e-4)
#define O_MAX_SCALE_VALUE (1e6)
#define O_CG_BEST_TOL (1e-12)
#define O_CG_TOL_FACTOR (0.2)
#define O_CG_RATE (1.5)
#define O_MIN_NORMALIZATION_FACTOR (1e-4)
#define O_MAX_NORMALIZATION_FACTOR (1e4)
#define O_NUM_RUIZ_PASSES (25)
#define O_NUM_L2_PASSES (1)
#define O_AA_MAX_WEIGHT_NORM (1e10)
#define O_AA_SAFEGUARD_FACTOR (1.0)

/* ---- linalg ---- */
scs_float o_dot(const scs_float *x, const scs_float *y, scs_int n);
scs_float o_norm_inf(const scs_float *x, scs_int n);
scs_float o_norm_2(const scs_float *x, scs_int n);
void o_axpy(scs_float *y, const scs_float *x, scs_float a, scs_int n); /* y += a x */
void o_scale(scs_float *x, scs_float a, scs_int n);
void o_accum_by_a(const ScsMatrix *A, const scs_float *x, scs_float *y);      /* y += A x  */
void o_accum_by_atrans(const ScsMatrix *A, const scs_float *x, scs_float *y); /* y += A'x  */
void o_accum_by_p(const ScsMatrix *P, const scs_float *x, scs_float *y);      /* y += P x (P upper tri, symmetric) */

/* ---- cones ---- */
typedef struct {
  ScsCone k;            /* deep copy (bu/bl get normalised in place) */
  scs_int m;            /* total rows */
  scs_int *boundaries;  /* block lengths, first = z+l+bsize (separately scalable rows) */
  scs_int n_boundaries;
  scs_float *s;         /* workspace m */
  scs_float box_t_warm; /* warm start for box cone t */
  /* PSD workspace */
  scs_float *Xs, *Vs, *es;
  scs_int max_s;
} OConeWork;

scs_int o_cone_dims(const ScsCone *k);
scs_int o_validate_cone(const ScsCone *k);
OConeWork *o_init_cone(const ScsCone *k, scs_int m);
void o_free_cone(OConeWork *c);
/* in-place Euclidean projection of x onto the PRIMAL cone K */
scs_int o_proj_cone(scs_float *x, OConeWork *c, const scs_float *r_y);
/* in-place projection onto the DUAL cone K* via Moreau (what ADMM uses) */
scs_int o_proj_dual_cone(scs_float *x, OConeWork *c, const scs_float *r_y);
void o_set_r_y(const OConeWork *c, scs_float scale, scs_float *r_y);
void o_enforce_cone_boundaries(const OConeWork *c, scs_float *vec, int use_mean);
void o_proj_exp_cone(scs_float *v, int primal);
void o_proj_power_cone(scs_float *v, scs_float a);
void o_proj_soc(scs_float *x, scs_int q);
scs_int o_proj_psd(scs_float *X, scs_int n, OConeWork *c);
scs_int o_proj_cpsd(scs_float *X, scs_int n, OConeWork *c);
/* symmetric eigen-decomposition (cyclic Jacobi), A n*n col-major in, eigvecs in V, eigvals in e */
void o_sym_eig(scs_float *A, scs_int n, scs_float *V, scs_float *e);

/* ---- normalisation ---- */
typedef struct {
  scs_float *D, *E; /* row (m) / col (n) scalings */
  scs_int m, n;
  scs_float primal_scale, dual_scale;
} OScaling;
OScaling *o_normalize_a_p(ScsMatrix *P, ScsMatrix *A, OConeWork *cone);
void o_normalize_b_c(OScaling *scal, scs_float *b, scs_float *c);
void o_normalize_sol(const OScaling *scal, ScsSolution *sol);
void o_un_normalize_sol(const OScaling *scal, ScsSolution *sol);
void o_free_scaling(OScaling *s);

/* ---- linear system ---- */
typedef struct OLinSys OLinSys;
OLinSys *o_init_lin_sys(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, int indirect);
void o_update_lin_sys_diag_r(OLinSys *p, const scs_float *diag_r);
/* solves [[R_x+P, A'],[A, -R_y]] z = b in place; s = warm start (indirect only) */
scs_int o_solve_lin_sys(OLinSys *p, scs_float *b, const scs_float *s, scs_float tol);
void o_free_lin_sys(OLinSys *p);
long o_lin_sys_cg_iters(const OLinSys *p);
long o_lin_sys_nnz_l(const OLinSys *p);
long o_lin_sys_symbolic(const ScsMatrix *A, const ScsMatrix *P, long *etree_height);

/* ---- anderson acceleration ---- */
typedef struct OAa OAa;
OAa *o_aa_init(scs_int dim, scs_int mem, scs_int type1, scs_float regularization,
               scs_float relaxation, scs_float safeguard_factor, scs_float max_weight_norm);
scs_float o_aa_apply(scs_float *f, const scs_float *x, OAa *a);
scs_int o_aa_safeguard(scs_float *f_new, scs_float *x_new, OAa *a);
void o_aa_reset(OAa *a);
void o_aa_free(OAa *a);
void o_aa_get_stats(const OAa *a, ScsAaStats *st);
scs_int o_aa_last_gamma(const OAa *a, scs_float *gamma);

/* small dense solve with partial pivoting; returns numerical rank (n if ok, <n if singular) */
scs_int o_dense_solve(scs_float *M, scs_float *rhs, scs_int n);

void oscs_set_num_threads(int n);
#endif
