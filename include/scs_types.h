/*
 * scs_types.h — data carriers of the SCS C API, as seen through the reference's
 * CPython glue.
 *
 * The upstream header (scs_source/include/scs.h) is ABSENT from the reference
 * snapshot (R:.gitmodules:1-3, R:meson.build:8-10).  Every field below is
 * evidenced by the line of R:scs/scsobject.h that reads or writes it; the
 * struct LAYOUT is this repo's own (a maintainer re-compiles the glue against
 * this header).
 *
 * Numeric types: GPU builds of the reference are int32-only
 * (R:meson.build:172-174, R:legacy_setup.py:79-80) and double precision unless
 * SFLOAT (R:meson.build:157-159).  This library is scs_int=int32, scs_float=f64.
 */
#ifndef SCS_TYPES_H_GUARD
#define SCS_TYPES_H_GUARD

#ifdef __cplusplus
extern "C" {
#endif

typedef int scs_int;      /* R:meson.build:172-174 (int32 mandatory on GPU) */
typedef double scs_float; /* R:meson.build:157-159 (double unless SFLOAT)   */

/* exit flags, R:scs/py/__init__.py:16-25 */
#define SCS_INFEASIBLE_INACCURATE (-7)
#define SCS_UNBOUNDED_INACCURATE (-6)
#define SCS_SIGINT (-5)
#define SCS_FAILED (-4)
#define SCS_INDETERMINATE (-3)
#define SCS_INFEASIBLE (-2)
#define SCS_UNBOUNDED (-1)
#define SCS_UNFINISHED (0)
#define SCS_SOLVED (1)
#define SCS_SOLVED_INACCURATE (2)

/* Sparse matrix in CSC; P is upper-triangular CSC.
 * R:scs/scsobject.h:594-605 (A), :636-647 (P). */
typedef struct {
  scs_float *x; /* values, length p[n]            */
  scs_int *i;   /* row indices, sorted per column (R:scs/py/__init__.py:140-141) */
  scs_int *p;   /* column pointers, length n+1    */
  scs_int m;    /* rows                           */
  scs_int n;    /* cols                           */
} ScsMatrix;

/* Problem data.  R:scs/scsobject.h:455,557-683. */
typedef struct {
  scs_int m;    /* rows of A    */
  scs_int n;    /* cols of A    */
  ScsMatrix *A; /* m x n        */
  ScsMatrix *P; /* n x n upper triangular or NULL */
  scs_float *b; /* length m     */
  scs_float *c; /* length n     */
} ScsData;

/* Cone, rows ordered z,l,box,q,s,cs,ep,ed,p.  R:scs/scsobject.h:684-749. */
typedef struct {
  scs_int z;     /* zero cone rows (dual: free)             :688-704 */
  scs_int l;     /* nonnegative rows                         :705-708 */
  scs_float *bu; /* box upper bounds, length bsize-1         :710-713 */
  scs_float *bl; /* box lower bounds, length bsize-1         :714-717 */
  scs_int bsize; /* total box cone length (t,s) = len(bu)+1  :722-724 */
  scs_int *q;    /* SOC dims                                 :726     */
  scs_int qsize;
  scs_int *s;    /* PSD matrix orders (vec len k(k+1)/2)     :730     */
  scs_int ssize;
  scs_int *cs;   /* complex PSD orders (vec len k*k: H_jj, then sqrt2 Re, sqrt2 Im of H_ij, i>j, by column) :734-737 */
  scs_int cssize;
  scs_int ep;    /* primal exponential triples               :742     */
  scs_int ed;    /* dual exponential triples                 :746     */
  scs_float *p;  /* power cone exponents in [-1,1], <0 dual  :738     */
  scs_int psize;
} ScsCone;

/* Settings.  Keyword table R:scs/scsobject.h:467-495, parsed into these
 * fields at :537-551, bools at :796-800, warm_start :869. */
typedef struct {
  scs_int normalize;
  scs_float scale;
  scs_int adaptive_scale;
  scs_float rho_x;
  scs_int max_iters;
  scs_float eps_abs;
  scs_float eps_rel;
  scs_float eps_infeas;
  scs_float alpha;
  scs_float time_limit_secs;
  scs_int verbose;
  scs_int warm_start;
  scs_int acceleration_lookback;
  scs_int acceleration_interval;
  scs_int acceleration_type_1;
  scs_float acceleration_regularization;
  scs_float acceleration_relaxation;
  const char *write_data_filename;
  const char *log_csv_filename;
} ScsSettings;

/* Primal-dual solution.  R:scs/scsobject.h:875-890. */
typedef struct {
  scs_float *x; /* n */
  scs_float *y; /* m */
  scs_float *s; /* m */
} ScsSolution;

/* Anderson-acceleration diagnostics.  R:scs/scsobject.h:1096-1107. */
typedef struct {
  scs_int iter;
  scs_int n_accept;
  scs_int n_reject_lapack;
  scs_int n_reject_rank0;
  scs_int n_reject_nonfinite;
  scs_int n_reject_weight_cap;
  scs_int n_safeguard_reject;
  scs_int last_rank;
  scs_float last_aa_norm;
  scs_float last_regularization;
} ScsAaStats;

/* Solve information.  R:scs/scsobject.h:1073-1095. */
typedef struct {
  scs_int iter;
  char status[128];
  char lin_sys_solver[128];
  scs_int status_val;
  scs_int scale_updates;
  scs_float pobj;
  scs_float dobj;
  scs_float res_pri;
  scs_float res_dual;
  scs_float gap;
  scs_float res_infeas;
  scs_float res_unbdd_a;
  scs_float res_unbdd_p;
  scs_float comp_slack;
  scs_float setup_time;   /* ms */
  scs_float solve_time;   /* ms */
  scs_float scale;
  scs_float lin_sys_time; /* ms */
  scs_float cone_time;    /* ms */
  scs_float accel_time;   /* ms */
  scs_int rejected_accel_steps;
  scs_int accepted_accel_steps;
  ScsAaStats aa_stats;
  /* extension (not in the reference's dict): total CG iterations of the solve */
  scs_int cg_iters;
} ScsInfo;

#ifdef __cplusplus
}
#endif
#endif
