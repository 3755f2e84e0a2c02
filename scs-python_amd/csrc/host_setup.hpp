// host_setup.hpp — init-time (one-off, O(nnz)) host work: validation, cone
// bookkeeping, CSC->CSR, symmetric expansion of P and the O(m+n) b/c scaling
// (the matrix equilibration itself runs on the device: normalize_dev.hpp).
//
// Plays the role of scs_source/src/normalize.c + the matrix helpers of
// scs_source/linsys/scs_matrix.c (R:meson.build:192,200; absent).  Algorithm:
// SURVEY App. A.6 — 25 Ruiz (inf-norm) passes + 1 l2 pass on [P A'; A 0], row
// scalings constant inside every non-separable cone, clamped to [1e-4, 1e4];
// then b,c scaled by D,E and a scalar sigma.
//   A_hat = D A E,  P_hat = E P E,  b_hat = sigma D b,  c_hat = sigma E c
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/scs_hip.h"

namespace scship {

struct HostCsr {
  int rows = 0, cols = 0;
  std::vector<int> rowptr, col;
  std::vector<double> val;
};

// Deep copy of the cone (the glue frees its view right after scs_init, R:scs/scsobject.h:908)
struct HostCone {
  int z = 0, l = 0, bsize = 0, ep = 0, ed = 0;
  std::vector<double> bu, bl, p;
  std::vector<int> q, s, cs;
  std::vector<int> boundaries;  // [z+l+bsize, q..., s(s+1)/2..., cs^2..., 3 x (ep+ed+psize)]
  int m = 0;
  int off_box = 0, off_q = 0, off_s = 0, off_cs = 0, off_ep = 0, off_ed = 0, off_p = 0;
};

inline long sd_size(long s) { return s * (s + 1) / 2; }

inline bool build_cone(const ScsCone *k, HostCone &c) {
  if (k->z < 0 || k->l < 0 || k->bsize < 0 || k->ep < 0 || k->ed < 0) return false;
  if (k->qsize < 0 || k->ssize < 0 || k->psize < 0 || k->cssize < 0) return false;
  c.z = k->z; c.l = k->l; c.bsize = k->bsize; c.ep = k->ep; c.ed = k->ed;
  if (k->bsize > 1) {
    c.bu.assign(k->bu, k->bu + k->bsize - 1);
    c.bl.assign(k->bl, k->bl + k->bsize - 1);
    for (int i = 0; i < k->bsize - 1; ++i)
      if (c.bl[i] > c.bu[i]) return false;
  }
  if (k->qsize) c.q.assign(k->q, k->q + k->qsize);
  if (k->ssize) c.s.assign(k->s, k->s + k->ssize);
  if (k->cssize) c.cs.assign(k->cs, k->cs + k->cssize);
  if (k->psize) c.p.assign(k->p, k->p + k->psize);
  for (int q : c.q) if (q < 0) return false;
  for (int s : c.s) if (s < 0 || s > 8192) return false;   // (psd.hpp kPsdMaxH: 16 * 512)
  for (int s : c.cs) if (s < 0 || s > 4096) return false;  // projected through its 2k x 2k real embedding
  for (double p : c.p) if (!(p >= -1 && p <= 1)) return false;
  long cnt = (long)c.z + c.l;
  c.off_box = (int)cnt; cnt += c.bsize;
  c.off_q = (int)cnt; for (int q : c.q) cnt += q;
  c.off_s = (int)cnt; for (int s : c.s) cnt += sd_size(s);
  c.off_cs = (int)cnt; for (int s : c.cs) cnt += (long)s * s;
  c.off_ep = (int)cnt; cnt += 3L * c.ep;
  c.off_ed = (int)cnt; cnt += 3L * c.ed;
  c.off_p = (int)cnt; cnt += 3L * (long)c.p.size();
  if (cnt > 2000000000L) return false;
  c.m = (int)cnt;
  c.boundaries.clear();
  c.boundaries.push_back(c.z + c.l + c.bsize);
  for (int q : c.q) c.boundaries.push_back(q);
  for (int s : c.s) c.boundaries.push_back((int)sd_size(s));
  for (int s : c.cs) c.boundaries.push_back(s * s);
  for (int i = 0; i < c.ep + c.ed + (int)c.p.size(); ++i) c.boundaries.push_back(3);
  return true;
}

inline bool validate_matrix(const ScsMatrix *A, int m, int n) {
  if (!A || !A->p || A->m != m || A->n != n) return false;
  if (A->p[0] != 0) return false;
  for (int j = 0; j < n; ++j) {
    if (A->p[j + 1] < A->p[j]) return false;
    for (int p = A->p[j]; p < A->p[j + 1]; ++p)
      if (A->i[p] < 0 || A->i[p] >= m) return false;
  }
  return true;
}

// CSC (m x n) -> CSR (m x n); stable counting sort keeps columns ascending inside a row
inline void csc_to_csr(int m, int n, const int *cp, const int *ri, const double *x, HostCsr &out) {
  const long nnz = cp[n];
  out.rows = m; out.cols = n;
  out.rowptr.assign(m + 1, 0);
  out.col.resize(nnz);
  out.val.resize(nnz);
  for (long p = 0; p < nnz; ++p) out.rowptr[ri[p] + 1]++;
  for (int i = 0; i < m; ++i) out.rowptr[i + 1] += out.rowptr[i];
  std::vector<int> next(out.rowptr.begin(), out.rowptr.end() - 1);
  for (int j = 0; j < n; ++j)
    for (int p = cp[j]; p < cp[j + 1]; ++p) {
      const int q = next[ri[p]]++;
      out.col[q] = j;
      out.val[q] = x[p];
    }
}

// upper-triangular CSC P -> full symmetric CSR (n x n), plus its diagonal
inline void sym_expand(int n, const int *cp, const int *ri, const double *x, HostCsr &out, std::vector<double> &diag) {
  out.rows = out.cols = n;
  out.rowptr.assign(n + 1, 0);
  diag.assign(n, 0.0);
  for (int j = 0; j < n; ++j)
    for (int p = cp[j]; p < cp[j + 1]; ++p) {
      const int i = ri[p];
      if (i > j) continue;
      out.rowptr[i + 1]++;
      if (i != j) out.rowptr[j + 1]++;
      else diag[j] += x[p];
    }
  for (int i = 0; i < n; ++i) out.rowptr[i + 1] += out.rowptr[i];
  out.col.resize(out.rowptr[n]);
  out.val.resize(out.rowptr[n]);
  std::vector<int> next(out.rowptr.begin(), out.rowptr.end() - 1);
  // two ordered passes keep every row's columns ascending: first the transposed
  // (strictly-lower) entries (col = i < row = j), then the upper entries (col = j >= row = i)
  for (int j = 0; j < n; ++j)  // lower part of row j: entries (i,j) with i<j, visited with i ascending
    for (int p = cp[j]; p < cp[j + 1]; ++p) {
      const int i = ri[p];
      if (i < j) { const int q = next[j]++; out.col[q] = i; out.val[q] = x[p]; }
    }
  for (int j = 0; j < n; ++j)  // upper part: row i gets column j, j ascending
    for (int p = cp[j]; p < cp[j + 1]; ++p) {
      const int i = ri[p];
      if (i <= j) { const int q = next[i]++; out.col[q] = j; out.val[q] = x[p]; }
    }
}

struct HostScaling {
  std::vector<double> D, E;
  double sigma = 1.0;
};

inline double apply_limit(double x) {
  x = x < 1e-4 ? 1.0 : x;
  return x > 1e4 ? 1e4 : x;
}
inline double safediv_pos(double x, double y) { return y < 1e-18 ? x / 1e-18 : x / y; }

inline void normalize_b_c(HostScaling &sc, double *b, int m, double *c, int n) {
  double nb = 0., nc = 0.;
  for (int i = 0; i < n; ++i) { c[i] *= sc.E[i]; nc = std::max(nc, std::fabs(c[i])); }
  for (int i = 0; i < m; ++i) { b[i] *= sc.D[i]; nb = std::max(nb, std::fabs(b[i])); }
  double sigma = std::max(nc, nb);
  sigma = sigma < 1e-4 ? 1.0 : sigma;
  sigma = sigma > 1e4 ? 1e4 : sigma;
  sigma = safediv_pos(1.0, sigma);
  for (int i = 0; i < n; ++i) c[i] *= sigma;
  for (int i = 0; i < m; ++i) b[i] *= sigma;
  sc.sigma = sigma;
}

}  // namespace scship
