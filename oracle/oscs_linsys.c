/*
 * oscs_linsys.c — ORACLE (test infrastructure): the two CPU linear-system
 * backends the parity story needs.
 *
 *  (1) DIRECT — sparse LDL' of the quasi-definite KKT matrix
 *          K = [[R_x + P, A'], [A, -R_y]]
 *      restating the role of scs_source/linsys/cpu/direct/private.c +
 *      external/qdldl (R:meson.build:238-256): fill-reducing ordering, an
 *      elimination-tree symbolic phase and an up-looking numeric phase — the
 *      published algorithm of Davis, "Algorithm 849: a concise sparse Cholesky
 *      factorization package" (2005), which QDLDL (Stellato et al., OSQP 2020)
 *      restates.  The ordering here is a plain greedy minimum-degree on the
 *      explicit elimination graph (the reference uses AMD, R:meson.build:219-234;
 *      any fill-reducing ordering yields the same solution).
 *  (2) INDIRECT — Jacobi-preconditioned CG on
 *          (R_x + P + A' R_y^{-1} A) x = r_x + A' R_y^{-1} r_y,   y = R_y^{-1}(A x - r_y)
 *      restating scs_source/linsys/cpu/indirect/private.c (R:meson.build:258-270,
 *      `-DINDIRECT=1`); SURVEY App. A.4.
 */
#include "oscs.h"

struct OLinSys {
  int indirect;
  scs_int n, m;
  const ScsMatrix *A, *P; /* borrowed (owned by the workspace) */
  const scs_float *diag_r;
  /* ---- direct ---- */
  scs_int N;             /* n + m */
  scs_int *Kp, *Ki;      /* permuted upper-triangular KKT, CSC */
  scs_float *Kx;
  scs_int *diag_idx;     /* position in Kx of diagonal entry of ORIGINAL index i */
  scs_int *perm, *iperm; /* perm[new] = old */
  scs_int *Lp, *Li, *Parent, *Lnz, *Flag, *Pattern;
  scs_float *Lx, *D, *Y, *bp;
  scs_float *diag_base;  /* P_ii for i<n, 0 for y rows */
  /* ---- indirect ---- */
  scs_float *p, *r, *Gp, *z, *M, *tmp;
  long tot_cg_its;
  /* all-core timing variant only (OSCS_OMP): CSR copy of A for the row-parallel A x */
  scs_int *csr_p, *csr_j;
  scs_float *csr_x;
};

/* ------------------------------------------------------------------ MD  */
typedef struct { scs_int *v; scs_int len, cap; } ivec;
static void ivec_push(ivec *a, scs_int x) {
  if (a->len == a->cap) { a->cap = a->cap ? 2 * a->cap : 8; a->v = (scs_int *)realloc(a->v, a->cap * sizeof(scs_int)); }
  a->v[a->len++] = x;
}
static int cmp_int(const void *a, const void *b) { return (*(const scs_int *)a > *(const scs_int *)b) - (*(const scs_int *)a < *(const scs_int *)b); }

/* greedy minimum degree on the explicit elimination graph */
static void min_degree_order(scs_int N, const scs_int *Cp, const scs_int *Ci, scs_int *perm) {
  ivec *adj = (ivec *)calloc(N, sizeof(ivec));
  char *gone = (char *)calloc(N, 1);
  scs_int *mark = (scs_int *)malloc(N * sizeof(scs_int));
  scs_int *tmp = (scs_int *)malloc(N * sizeof(scs_int));
  scs_int j, p, k;
  for (j = 0; j < N; ++j) mark[j] = -1;
  for (j = 0; j < N; ++j)
    for (p = Cp[j]; p < Cp[j + 1]; ++p) {
      scs_int i = Ci[p];
      if (i != j) { ivec_push(&adj[i], j); ivec_push(&adj[j], i); }
    }
  for (j = 0; j < N; ++j) { /* dedupe */
    qsort(adj[j].v, adj[j].len, sizeof(scs_int), cmp_int);
    scs_int w = 0;
    for (p = 0; p < adj[j].len; ++p)
      if (p == 0 || adj[j].v[p] != adj[j].v[p - 1]) adj[j].v[w++] = adj[j].v[p];
    adj[j].len = w;
  }
  for (k = 0; k < N; ++k) {
    scs_int best = -1, bestdeg = N + 1;
    for (j = 0; j < N; ++j)
      if (!gone[j] && adj[j].len < bestdeg) { bestdeg = adj[j].len; best = j; }
    perm[k] = best;
    gone[best] = 1;
    /* clique among neighbours of best */
    scs_int nn = adj[best].len;
    for (p = 0; p < nn; ++p) {
      scs_int u = adj[best].v[p];
      /* new adj[u] = (adj[u] U adj[best]) \ {u, best} */
      scs_int w = 0, q;
      for (q = 0; q < adj[u].len; ++q) {
        scs_int x = adj[u].v[q];
        if (x != best && mark[x] != u) { mark[x] = u; tmp[w++] = x; }
      }
      for (q = 0; q < nn; ++q) {
        scs_int x = adj[best].v[q];
        if (x != u && mark[x] != u) { mark[x] = u; tmp[w++] = x; }
      }
      if (w > adj[u].cap) { adj[u].cap = w + 8; adj[u].v = (scs_int *)realloc(adj[u].v, adj[u].cap * sizeof(scs_int)); }
      memcpy(adj[u].v, tmp, w * sizeof(scs_int));
      adj[u].len = w;
    }
    free(adj[best].v);
    adj[best].v = NULL;
    adj[best].len = adj[best].cap = 0;
    /* marks are keyed by the surviving neighbour u: clear them before u can recur */
    for (p = 0; p < N; ++p) mark[p] = -1;
  }
  for (j = 0; j < N; ++j) free(adj[j].v);
  free(adj); free(gone); free(mark); free(tmp);
}

/* ------------------------------------------------------------------ LDL */
static void ldl_symbolic(scs_int n, const scs_int *Ap, const scs_int *Ai, scs_int *Lp, scs_int *Parent,
                         scs_int *Lnz, scs_int *Flag) {
  scs_int i, k, p;
  for (k = 0; k < n; k++) {
    Parent[k] = -1;
    Flag[k] = k;
    Lnz[k] = 0;
    for (p = Ap[k]; p < Ap[k + 1]; p++) {
      i = Ai[p];
      if (i < k) {
        for (; Flag[i] != k; i = Parent[i]) {
          if (Parent[i] == -1) Parent[i] = k;
          Lnz[i]++;
          Flag[i] = k;
        }
      }
    }
  }
  Lp[0] = 0;
  for (k = 0; k < n; k++) Lp[k + 1] = Lp[k] + Lnz[k];
}

static scs_int ldl_numeric(scs_int n, const scs_int *Ap, const scs_int *Ai, const scs_float *Ax,
                           const scs_int *Lp, const scs_int *Parent, scs_int *Lnz, scs_int *Li,
                           scs_float *Lx, scs_float *D, scs_float *Y, scs_int *Pattern, scs_int *Flag) {
  scs_float yi, l_ki;
  scs_int i, k, p, p2, len, top;
  for (k = 0; k < n; k++) {
    Y[k] = 0.0;
    top = n;
    Flag[k] = k;
    Lnz[k] = 0;
    for (p = Ap[k]; p < Ap[k + 1]; p++) {
      i = Ai[p];
      if (i <= k) {
        Y[i] += Ax[p];
        for (len = 0; Flag[i] != k; i = Parent[i]) {
          Pattern[len++] = i;
          Flag[i] = k;
        }
        while (len > 0) Pattern[--top] = Pattern[--len];
      }
    }
    D[k] = Y[k];
    Y[k] = 0.0;
    for (; top < n; top++) {
      i = Pattern[top];
      yi = Y[i];
      Y[i] = 0.0;
      p2 = Lp[i] + Lnz[i];
      for (p = Lp[i]; p < p2; p++) Y[Li[p]] -= Lx[p] * yi;
      l_ki = yi / D[i];
      D[k] -= l_ki * yi;
      Li[p] = k;
      Lx[p] = l_ki;
      Lnz[i]++;
    }
    if (D[k] == 0.0) return k;
  }
  return n;
}

static void ldl_solve(scs_int n, scs_float *X, const scs_int *Lp, const scs_int *Li, const scs_float *Lx,
                      const scs_float *D) {
  scs_int j, p;
  for (j = 0; j < n; j++)
    for (p = Lp[j]; p < Lp[j + 1]; p++) X[Li[p]] -= Lx[p] * X[j];
  for (j = 0; j < n; j++) X[j] /= D[j];
  for (j = n - 1; j >= 0; j--)
    for (p = Lp[j]; p < Lp[j + 1]; p++) X[j] -= Lx[p] * X[Li[p]];
}

/* assemble upper-tri KKT in ORIGINAL ordering as triplets, permute, compress */
static void build_kkt(OLinSys *w) {
  const ScsMatrix *A = w->A, *P = w->P;
  scs_int n = w->n, m = w->m, N = n + m, j, p, k;
  scs_int nnz_max = N + A->p[n] + (P ? P->p[n] : 0);
  scs_int *ti = (scs_int *)malloc(nnz_max * sizeof(scs_int));
  scs_int *tj = (scs_int *)malloc(nnz_max * sizeof(scs_int));
  scs_float *tx = (scs_float *)malloc(nnz_max * sizeof(scs_float));
  scs_int *tdiag = (scs_int *)malloc(nnz_max * sizeof(scs_int)); /* original diag index or -1 */
  scs_int nz = 0;
  w->diag_base = (scs_float *)calloc(N, sizeof(scs_float));
  if (P)
    for (j = 0; j < n; ++j)
      for (p = P->p[j]; p < P->p[j + 1]; ++p) {
        scs_int i = P->i[p];
        if (i > j) continue;
        if (i == j) { w->diag_base[j] += P->x[p]; continue; }
        ti[nz] = i; tj[nz] = j; tx[nz] = P->x[p]; tdiag[nz] = -1; nz++;
      }
  for (j = 0; j < N; ++j) { ti[nz] = j; tj[nz] = j; tx[nz] = 0.; tdiag[nz] = j; nz++; }
  for (j = 0; j < n; ++j) /* A' block: entry (col j of x-block, row n+i) -> upper tri (j, n+i) */
    for (p = A->p[j]; p < A->p[j + 1]; ++p) {
      ti[nz] = j; tj[nz] = n + A->i[p]; tx[nz] = A->x[p]; tdiag[nz] = -1; nz++;
    }
  /* ordering on the pattern in original order (build a temporary CSC) */
  {
    scs_int *Cp = (scs_int *)calloc(N + 1, sizeof(scs_int));
    scs_int *Ci = (scs_int *)malloc(nz * sizeof(scs_int));
    scs_int *cnt = (scs_int *)calloc(N, sizeof(scs_int));
    for (k = 0; k < nz; ++k) Cp[tj[k] + 1]++;
    for (j = 0; j < N; ++j) Cp[j + 1] += Cp[j];
    for (k = 0; k < nz; ++k) Ci[Cp[tj[k]] + cnt[tj[k]]++] = ti[k];
    w->perm = (scs_int *)malloc(N * sizeof(scs_int));
    w->iperm = (scs_int *)malloc(N * sizeof(scs_int));
    min_degree_order(N, Cp, Ci, w->perm);
    for (j = 0; j < N; ++j) w->iperm[w->perm[j]] = j;
    free(Cp); free(Ci); free(cnt);
  }
  /* permute: (i,j) -> (iperm[i], iperm[j]) kept in the upper triangle */
  {
    scs_int *cnt = (scs_int *)calloc(N, sizeof(scs_int));
    w->Kp = (scs_int *)calloc(N + 1, sizeof(scs_int));
    w->Ki = (scs_int *)malloc(nz * sizeof(scs_int));
    w->Kx = (scs_float *)malloc(nz * sizeof(scs_float));
    w->diag_idx = (scs_int *)malloc(N * sizeof(scs_int));
    for (k = 0; k < nz; ++k) {
      scs_int a = w->iperm[ti[k]], b = w->iperm[tj[k]];
      if (a > b) { scs_int t = a; a = b; b = t; }
      ti[k] = a; tj[k] = b;
      w->Kp[b + 1]++;
    }
    for (j = 0; j < N; ++j) w->Kp[j + 1] += w->Kp[j];
    for (k = 0; k < nz; ++k) {
      scs_int pos = w->Kp[tj[k]] + cnt[tj[k]]++;
      w->Ki[pos] = ti[k];
      w->Kx[pos] = tx[k];
      if (tdiag[k] >= 0) w->diag_idx[tdiag[k]] = pos;
    }
    free(cnt);
  }
  free(ti); free(tj); free(tx); free(tdiag);
}

static scs_int factorize(OLinSys *w) {
  scs_int N = w->N, i;
  for (i = 0; i < w->n; ++i) w->Kx[w->diag_idx[i]] = w->diag_base[i] + w->diag_r[i];
  for (i = w->n; i < N; ++i) w->Kx[w->diag_idx[i]] = -w->diag_r[i];
  return ldl_numeric(N, w->Kp, w->Ki, w->Kx, w->Lp, w->Parent, w->Lnz, w->Li, w->Lx, w->D, w->Y,
                     w->Pattern, w->Flag) == N ? 0 : -1;
}

/* ------------------------------------------------------------- indirect */
static void set_preconditioner(OLinSys *w) {
  const ScsMatrix *A = w->A, *P = w->P;
  scs_int j, p;
  for (j = 0; j < w->n; ++j) {
    scs_float d = w->diag_r[j];
    for (p = A->p[j]; p < A->p[j + 1]; ++p) d += A->x[p] * A->x[p] / w->diag_r[w->n + A->i[p]];
    if (P)
      for (p = P->p[j]; p < P->p[j + 1]; ++p)
        if (P->i[p] == j) d += P->x[p];
    w->M[j] = 1.0 / d;
  }
}

/* y = (R_x + P + A' R_y^{-1} A) x */
static void mat_vec(OLinSys *w, const scs_float *x, scs_float *y) {
  scs_int i;
  scs_float *z = w->tmp;
  memset(y, 0, w->n * sizeof(scs_float));
  if (w->P) o_accum_by_p(w->P, x, y);
#ifdef OSCS_OMP
  /* all-core timing variant: A x by rows over a CSR copy (each row summed in ascending column order, as the CSC
   * scatter below does) */
  O_PAR_FOR(w->m)
  for (i = 0; i < w->m; ++i) {
    scs_float acc = 0.;
    for (scs_int q = w->csr_p[i]; q < w->csr_p[i + 1]; ++q) acc += w->csr_x[q] * x[w->csr_j[q]];
    z[i] = acc / w->diag_r[w->n + i];
  }
#else
  memset(z, 0, w->m * sizeof(scs_float));
  o_accum_by_a(w->A, x, z);
  for (i = 0; i < w->m; ++i) z[i] /= w->diag_r[w->n + i];
#endif
  o_accum_by_atrans(w->A, z, y);
  O_PAR_FOR(w->n)
  for (i = 0; i < w->n; ++i) y[i] += w->diag_r[i] * x[i];
}

static scs_int pcg(OLinSys *w, const scs_float *s, scs_float *b, scs_int max_its, scs_float tol) {
  scs_int i, j, n = w->n;
  scs_float ztr, ztr_prev, alpha;
  scs_float *p = w->p, *Gp = w->Gp, *r = w->r, *z = w->z, *M = w->M;
  if (!s) {
    memcpy(r, b, n * sizeof(scs_float));
    memset(b, 0, n * sizeof(scs_float));
  } else {
    mat_vec(w, s, r);
    for (j = 0; j < n; ++j) r[j] = b[j] - r[j];
    memcpy(b, s, n * sizeof(scs_float));
  }
  if (o_norm_inf(r, n) < OMAX(tol, 1e-12)) return 0;
  O_PAR_FOR(n)
  for (j = 0; j < n; ++j) z[j] = M[j] * r[j];
  ztr = o_dot(z, r, n);
  memcpy(p, z, n * sizeof(scs_float));
  for (i = 0; i < max_its; ++i) {
    mat_vec(w, p, Gp);
    alpha = ztr / o_dot(p, Gp, n);
    o_axpy(b, p, alpha, n);
    o_axpy(r, Gp, -alpha, n);
    if (o_norm_inf(r, n) < tol) return i + 1;
    O_PAR_FOR(n)
    for (j = 0; j < n; ++j) z[j] = M[j] * r[j];
    ztr_prev = ztr;
    ztr = o_dot(z, r, n);
    o_scale(p, ztr / ztr_prev, n);
    o_axpy(p, z, 1., n);
  }
  return i;
}

/* ------------------------------------------------------------------ API */
OLinSys *o_init_lin_sys(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, int indirect) {
  OLinSys *w = (OLinSys *)calloc(1, sizeof(OLinSys));
  w->indirect = indirect;
  w->n = A->n; w->m = A->m; w->N = A->n + A->m;
  w->A = A; w->P = P; w->diag_r = diag_r;
  if (indirect) {
    w->p = (scs_float *)calloc(w->n, sizeof(scs_float));
    w->r = (scs_float *)calloc(w->n, sizeof(scs_float));
    w->Gp = (scs_float *)calloc(w->n, sizeof(scs_float));
    w->z = (scs_float *)calloc(w->n, sizeof(scs_float));
    w->M = (scs_float *)calloc(w->n, sizeof(scs_float));
    w->tmp = (scs_float *)calloc(w->m, sizeof(scs_float));
    set_preconditioner(w);
#ifdef OSCS_OMP
    {  /* CSR copy of (the current, equilibrated) A */
      const scs_int nnz = A->p[A->n];
      scs_int j, q, *cur;
      w->csr_p = (scs_int *)calloc(w->m + 1, sizeof(scs_int));
      w->csr_j = (scs_int *)malloc(OMAX(nnz, 1) * sizeof(scs_int));
      w->csr_x = (scs_float *)malloc(OMAX(nnz, 1) * sizeof(scs_float));
      for (q = 0; q < nnz; ++q) w->csr_p[A->i[q] + 1]++;
      for (j = 0; j < w->m; ++j) w->csr_p[j + 1] += w->csr_p[j];
      cur = (scs_int *)malloc(OMAX(w->m, 1) * sizeof(scs_int));
      memcpy(cur, w->csr_p, w->m * sizeof(scs_int));
      for (j = 0; j < A->n; ++j)
        for (q = A->p[j]; q < A->p[j + 1]; ++q) {
          const scs_int dst = cur[A->i[q]]++;
          w->csr_j[dst] = j;
          w->csr_x[dst] = A->x[q];
        }
      free(cur);
    }
#endif
    return w;
  }
  build_kkt(w);
  scs_int N = w->N;
  w->Lp = (scs_int *)calloc(N + 1, sizeof(scs_int));
  w->Parent = (scs_int *)calloc(N, sizeof(scs_int));
  w->Lnz = (scs_int *)calloc(N, sizeof(scs_int));
  w->Flag = (scs_int *)calloc(N, sizeof(scs_int));
  w->Pattern = (scs_int *)calloc(N, sizeof(scs_int));
  w->D = (scs_float *)calloc(N, sizeof(scs_float));
  w->Y = (scs_float *)calloc(N, sizeof(scs_float));
  w->bp = (scs_float *)calloc(N, sizeof(scs_float));
  ldl_symbolic(N, w->Kp, w->Ki, w->Lp, w->Parent, w->Lnz, w->Flag);
  w->Li = (scs_int *)malloc(OMAX(w->Lp[N], 1) * sizeof(scs_int));
  w->Lx = (scs_float *)malloc(OMAX(w->Lp[N], 1) * sizeof(scs_float));
  if (factorize(w) < 0) { o_free_lin_sys(w); return NULL; }
  return w;
}

void o_update_lin_sys_diag_r(OLinSys *w, const scs_float *diag_r) {
  w->diag_r = diag_r;
  if (w->indirect) set_preconditioner(w);
  else factorize(w);
}

scs_int o_solve_lin_sys(OLinSys *w, scs_float *b, const scs_float *s, scs_float tol) {
  scs_int i, n = w->n, m = w->m;
  if (!w->indirect) {
    for (i = 0; i < w->N; ++i) w->bp[i] = b[w->perm[i]];
    ldl_solve(w->N, w->bp, w->Lp, w->Li, w->Lx, w->D);
    for (i = 0; i < w->N; ++i) b[w->perm[i]] = w->bp[i];
    return 0;
  }
  if (o_norm_inf(b, n + m) <= 1e-12) { memset(b, 0, (n + m) * sizeof(scs_float)); return 0; }
  /* b[:n] = rx + A' R_y^{-1} ry */
  for (i = 0; i < m; ++i) w->tmp[i] = b[n + i] / w->diag_r[n + i];
  o_accum_by_atrans(w->A, w->tmp, b);
  w->tot_cg_its += pcg(w, s, b, 10 * n, tol);
  /* y = R_y^{-1} (A x - ry) */
  o_scale(&b[n], -1., m);
  o_accum_by_a(w->A, b, &b[n]);
  for (i = 0; i < m; ++i) b[n + i] /= w->diag_r[n + i];
  return 0;
}

long o_lin_sys_cg_iters(const OLinSys *w) { return w->tot_cg_its; }
long o_lin_sys_nnz_l(const OLinSys *w) { return w->indirect ? 0 : (long)w->Lp[w->N]; }

void o_free_lin_sys(OLinSys *w) {
  if (!w) return;
  free(w->Kp); free(w->Ki); free(w->Kx); free(w->diag_idx); free(w->perm); free(w->iperm);
  free(w->Lp); free(w->Li); free(w->Parent); free(w->Lnz); free(w->Flag); free(w->Pattern);
  free(w->Lx); free(w->D); free(w->Y); free(w->bp); free(w->diag_base);
  free(w->p); free(w->r); free(w->Gp); free(w->z); free(w->M); free(w->tmp);
  free(w->csr_p); free(w->csr_j); free(w->csr_x);
  free(w);
}
