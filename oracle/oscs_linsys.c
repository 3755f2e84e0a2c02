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

/* Approximate minimum degree on the QUOTIENT graph (round 4; the reference orders its KKT matrix with SuiteSparse AMD before
 * QDLDL, R:meson.build:219-234,242 — sources absent; this restates the published algorithm: Amestoy, Davis, Duff 1996, without
 * mass elimination of indistinguishable variables).  An eliminated pivot p becomes an ELEMENT whose list L_p holds the variables
 * its clique connects; the clique is never formed.  A variable i keeps A_i (adjacent variables, pruned of everything an element
 * already connects it to) and E_i (adjacent elements); its degree is bounded by
 *     d_i = min(n - k - 1,  d_i + |L_p \ i|,  |A_i| + |L_p \ i| + sum_{e in E_i, e != p} |L_e \ L_p|),
 * with |L_e \ L_p| from one pass over the elements adjacent to L_p.  Elements inside L_p are absorbed.  Any permutation is a valid
 * ordering: a weaker bound only costs fill.  O(nnz) memory, near-linear time — the greedy explicit-clique version above took
 * 37 s at N = 12 000 and capped the LDL' ladder of bench.py at m = 4000.  OSCS_ORDER=md selects the old one. */
static void amd_order(scs_int N, const scs_int *Cp, const scs_int *Ci, scs_int *perm) {
  ivec *A = (ivec *)calloc(N, sizeof(ivec)), *E = (ivec *)calloc(N, sizeof(ivec)), *L = (ivec *)calloc(N, sizeof(ivec));
  char *st = (char *)calloc(N, 1);                    /* 0 variable, 1 live element, 2 absorbed element */
  scs_int *deg = (scs_int *)malloc(N * sizeof(scs_int)), *mark = (scs_int *)malloc(N * sizeof(scs_int));
  scs_int *w = (scs_int *)malloc(N * sizeof(scs_int)), *wst = (scs_int *)malloc(N * sizeof(scs_int));
  scs_int *head = (scs_int *)malloc((N + 1) * sizeof(scs_int)), *next = (scs_int *)malloc(N * sizeof(scs_int)),
          *prev = (scs_int *)malloc(N * sizeof(scs_int));
  scs_int j, p, k, mindeg = 0;
  for (j = 0; j < N; ++j) { mark[j] = -1; wst[j] = -1; }
  for (j = 0; j <= N; ++j) head[j] = -1;
  for (j = 0; j < N; ++j)
    for (p = Cp[j]; p < Cp[j + 1]; ++p) {
      scs_int i = Ci[p];
      if (i != j) { ivec_push(&A[i], j); ivec_push(&A[j], i); }
    }
  for (j = 0; j < N; ++j) { /* dedupe */
    scs_int wr = 0;
    qsort(A[j].v, A[j].len, sizeof(scs_int), cmp_int);
    for (p = 0; p < A[j].len; ++p)
      if (p == 0 || A[j].v[p] != A[j].v[p - 1]) A[j].v[wr++] = A[j].v[p];
    A[j].len = wr;
    deg[j] = wr;
  }
#define AMD_INSERT(i) do { scs_int d_ = deg[i]; next[i] = head[d_]; prev[i] = -1; if (head[d_] >= 0) prev[head[d_]] = (i); head[d_] = (i); } while (0)
#define AMD_REMOVE(i) do { scs_int d_ = deg[i]; if (prev[i] >= 0) next[prev[i]] = next[i]; else head[d_] = next[i]; if (next[i] >= 0) prev[next[i]] = prev[i]; } while (0)
  for (j = 0; j < N; ++j) AMD_INSERT(j);
  for (k = 0; k < N; ++k) {
    scs_int piv, q, nl;
    while (mindeg < N && head[mindeg] < 0) ++mindeg;
    piv = head[mindeg];
    AMD_REMOVE(piv);
    perm[k] = piv;
    /* L_piv = (A_piv U union of L_e, e in E_piv) \ piv; the elements of E_piv are absorbed */
    L[piv].len = 0;
    mark[piv] = k;
    for (p = 0; p < A[piv].len; ++p) {
      scs_int v = A[piv].v[p];
      if (st[v] == 0 && mark[v] != k) { mark[v] = k; ivec_push(&L[piv], v); }
    }
    for (p = 0; p < E[piv].len; ++p) {
      scs_int e = E[piv].v[p];
      if (st[e] != 1) continue;
      for (q = 0; q < L[e].len; ++q) {
        scs_int v = L[e].v[q];
        if (st[v] == 0 && mark[v] != k) { mark[v] = k; ivec_push(&L[piv], v); }
      }
      st[e] = 2;
      free(L[e].v); L[e].v = NULL; L[e].len = L[e].cap = 0;
    }
    st[piv] = 1;
    free(A[piv].v); A[piv].v = NULL; A[piv].len = A[piv].cap = 0;
    free(E[piv].v); E[piv].v = NULL; E[piv].len = E[piv].cap = 0;
    nl = L[piv].len;
    /* clean E_i of absorbed elements, then |L_e \ L_piv| for every element next to L_piv */
    for (p = 0; p < nl; ++p) {
      scs_int i = L[piv].v[p], wr = 0;
      for (q = 0; q < E[i].len; ++q) {
        scs_int e = E[i].v[q];
        if (st[e] == 1) E[i].v[wr++] = e;
      }
      E[i].len = wr;
      for (q = 0; q < wr; ++q) {
        scs_int e = E[i].v[q];
        if (wst[e] != k) { wst[e] = k; w[e] = L[e].len; }
        w[e]--;
      }
    }
    for (p = 0; p < nl; ++p) {
      scs_int i = L[piv].v[p], wr = 0, d;
      AMD_REMOVE(i);
      for (q = 0; q < A[i].len; ++q) { /* prune: everything in L_piv (and piv) is reachable through the new element */
        scs_int v = A[i].v[q];
        if (st[v] == 0 && mark[v] != k) A[i].v[wr++] = v;
      }
      A[i].len = wr;
      d = wr + (nl - 1);
      wr = 0;
      for (q = 0; q < E[i].len; ++q) {
        scs_int e = E[i].v[q];
        if (st[e] != 1) continue;
        if (w[e] == 0) { st[e] = 2; free(L[e].v); L[e].v = NULL; L[e].len = L[e].cap = 0; continue; } /* L_e inside L_piv: absorbed */
        E[i].v[wr++] = e;
        d += w[e];
      }
      E[i].len = wr;
      ivec_push(&E[i], piv);
      if (d > deg[i] + nl - 1) d = deg[i] + nl - 1;
      if (d > N - k - 2) d = N - k - 2;
      if (d < 0) d = 0;
      deg[i] = d;
      AMD_INSERT(i);
      if (d < mindeg) mindeg = d;
    }
  }
#undef AMD_INSERT
#undef AMD_REMOVE
  for (j = 0; j < N; ++j) { free(A[j].v); free(E[j].v); free(L[j].v); }
  free(A); free(E); free(L); free(st); free(deg); free(mark); free(w); free(wst); free(head); free(next); free(prev);
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
    {
      const char *eo = getenv("OSCS_ORDER");
      if (eo && eo[0] == 'm') min_degree_order(N, Cp, Ci, w->perm);
      else amd_order(N, Cp, Ci, w->perm);
    }
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
  o_par_zero(y, w->n);
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
    o_par_copy(r, b, n);
    o_par_zero(b, n);
  } else {
    mat_vec(w, s, r);
    for (j = 0; j < n; ++j) r[j] = b[j] - r[j];
    o_par_copy(b, s, n);
  }
  if (o_norm_inf(r, n) < OMAX(tol, 1e-12)) return 0;
  O_PAR_FOR(n)
  for (j = 0; j < n; ++j) z[j] = M[j] * r[j];
  ztr = o_dot(z, r, n);
  o_par_copy(p, z, n);
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
    w->p = o_vec_calloc(w->n);
    w->r = o_vec_calloc(w->n);
    w->Gp = o_vec_calloc(w->n);
    w->z = o_vec_calloc(w->n);
    w->M = o_vec_calloc(w->n);
    w->tmp = o_vec_calloc(w->m);
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
      {  /* first touch: the copy the row-parallel product streams is written by the threads that will read it */
        scs_int *cj = (scs_int *)malloc(OMAX(nnz, 1) * sizeof(scs_int));
        scs_float *cx = (scs_float *)malloc(OMAX(nnz, 1) * sizeof(scs_float));
        O_PAR_FOR(w->m)
        for (scs_int i = 0; i < w->m; ++i)
          for (scs_int t = w->csr_p[i]; t < w->csr_p[i + 1]; ++t) { cj[t] = w->csr_j[t]; cx[t] = w->csr_x[t]; }
        free(w->csr_j); free(w->csr_x);
        w->csr_j = cj; w->csr_x = cx;
      }
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

/* symbolic phase only (no values, no numeric factorisation): nnz(L) of the LDL' factor of the KKT pattern under the ordering above, and the
 * HEIGHT of its elimination tree — the length of the dependency chain any factorisation (CPU or device) has to walk column after column.
 * What DESIGN.md §7 prices a sparse direct solver on the device with (tools/ldl_fill_table.py). */
long o_lin_sys_symbolic(const ScsMatrix *A, const ScsMatrix *P, long *etree_height) {
  OLinSys *w = (OLinSys *)calloc(1, sizeof(OLinSys));
  long lnz;
  scs_int N, k;
  w->n = A->n; w->m = A->m; w->N = A->n + A->m;
  w->A = A; w->P = P;
  build_kkt(w);
  N = w->N;
  w->Lp = (scs_int *)calloc(N + 1, sizeof(scs_int));
  w->Parent = (scs_int *)calloc(N, sizeof(scs_int));
  w->Lnz = (scs_int *)calloc(N, sizeof(scs_int));
  w->Flag = (scs_int *)calloc(N, sizeof(scs_int));
  {  /* column counts in long arithmetic: nnz(L) of the larger replicas does not fit scs_int */
    scs_int i, p;
    lnz = 0;
    for (k = 0; k < N; k++) {
      w->Parent[k] = -1;
      w->Flag[k] = k;
      for (p = w->Kp[k]; p < w->Kp[k + 1]; p++) {
        i = w->Ki[p];
        if (i < k)
          for (; w->Flag[i] != k; i = w->Parent[i]) {
            if (w->Parent[i] == -1) w->Parent[i] = k;
            lnz++;
            w->Flag[i] = k;
          }
      }
    }
  }
  if (etree_height) {  /* Parent[k] > k: depth by one backward pass */
    long h = 0;
    scs_int *depth = (scs_int *)calloc(N, sizeof(scs_int));
    for (k = N - 1; k >= 0; --k) {
      depth[k] = w->Parent[k] >= 0 ? depth[w->Parent[k]] + 1 : 1;
      if (depth[k] > h) h = depth[k];
    }
    free(depth);
    *etree_height = h;
  }
  o_free_lin_sys(w);
  return lnz;
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
