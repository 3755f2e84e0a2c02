// psd.hpp — K9: batched projection onto the PSD cone.
//
// Plays the role of the LAPACK syev* path of scs_source/src/cones.c under
// USE_LAPACK (R:meson.build:145-147,188; absent).  Vector layout per cone:
// lower triangle, column-major, off-diagonals scaled by sqrt(2)
// (R:test/gen_random_cone_prob.py:153-173, R:test/test_scs_coverage.py:1387-1393).
//
// One workgroup per matrix (all matrices of the cone run concurrently, one CU
// each).  Eigen-decomposition: two-sided cyclic Jacobi in the round-robin
// (tournament) parallel ordering.  Each step applies n/2 disjoint rotations
// J = prod_k J_k:  A <- J' A J is done in ONE pass by giving every (k,k') pair of
// rotations its own 2x2 block of A (each element belongs to exactly one block, so
// the update is in place), V <- V J column-wise.  Two barriers per step.
// Reconstruction X+ = V diag(max(lambda,0)) V' is a dense contraction on the
// fp64 matrix cores (v_mfma_f64_16x16x4_f64) over 16x16 output tiles.
// The PSD cone is self-dual, so Pi_{K*} = Pi_K.
#pragma once
#include "common.hpp"

namespace scship {

constexpr int kPsdThreads = 1024;
constexpr int kPsdMaxSweeps = 40;

struct PsdBatch {
  const int *off;    // start of each cone's vector inside the m-vector slice
  const int *order;  // matrix order n_c
  const long *woff;  // offset (in doubles) of this matrix's scratch: A (npad*npad) then V (npad*npad) then lam (npad)
  int count;
};

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void rr_pair(int r, int k, int N, int &p, int &q) {
  // round-robin tournament on N (even) players, round r in [0, N-1)
  if (k == 0) { p = N - 1; q = r % (N - 1); }
  else { p = (r + k) % (N - 1); q = (r - k + (N - 1)) % (N - 1); }
  if (p > q) { const int t = p; p = q; q = t; }
}

__global__ __launch_bounds__(kPsdThreads) void k_proj_psd(double *x, PsdBatch B, double *scratch) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *cs = reinterpret_cast<double *>(smem_raw);  // [N/2] cos
  double *sn = cs + 512;                              // [N/2] sin
  double *red = sn + 512;                             // [16]
  double *bc = red + 16;                              // [2] broadcast
  const int cidx = blockIdx.x;
  const int n = B.order[cidx];
  double *X = x + B.off[cidx];
  const int tid = threadIdx.x;
  if (n == 0) return;
  if (n == 1) {
    if (tid == 0) X[0] = fmax(X[0], 0.);
    return;
  }
  const int N = (n + 1) & ~1;       // even number of players
  const int ld = (n + 15) & ~15;    // padded leading dimension (MFMA tiles)
  double *A = scratch + B.woff[cidx];
  double *V = A + (size_t)ld * ld;
  double *lam = V + (size_t)ld * ld;
  const double isq2 = 0.70710678118654752440, sq2 = 1.41421356237309504880;

  // ---- unpack (lower tri, col-major, off-diag / sqrt2), V = I, zero padding ----
  for (int e = tid; e < ld * ld; e += kPsdThreads) {
    const int i = e % ld, j = e / ld;
    A[e] = 0.;
    V[e] = (i == j && i < n) ? 1. : 0.;
  }
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    // column j of the packed vector starts at j*n - j(j-1)/2
    const long base = (long)j * n - (long)j * (j - 1) / 2;
    for (int i = j + tid; i < n; i += kPsdThreads) {
      double v = X[base + (i - j)];
      if (i != j) v *= isq2;
      A[i + (size_t)ld * j] = v;
      A[j + (size_t)ld * i] = v;
    }
  }
  __syncthreads();

  // ---- Jacobi sweeps ----
  for (int sweep = 0; sweep < kPsdMaxSweeps; ++sweep) {
    // convergence: off-diagonal mass vs total
    double off = 0., tot = 0.;
    for (int e = tid; e < n * n; e += kPsdThreads) {
      const int i = e % n, j = e / n;
      const double a = A[i + (size_t)ld * j];
      tot += a * a;
      if (i != j) off += a * a;
    }
    off = block_sum<kPsdThreads>(off, red);
    tot = block_sum<kPsdThreads>(tot, red);
    if (tid == 0) bc[0] = (off <= 1e-30 * tot || off == 0.) ? 1. : 0.;
    __syncthreads();
    const bool done = bc[0] != 0.;
    __syncthreads();
    if (done) break;

    for (int r = 0; r < N - 1; ++r) {
      // phase A: rotation angles of the N/2 disjoint pairs
      for (int k = tid; k < N / 2; k += kPsdThreads) {
        int p, q;
        rr_pair(r, k, N, p, q);
        double c = 1., s = 0.;
        if (q < n) {
          const double apq = A[p + (size_t)ld * q];
          if (fabs(apq) > 1e-300) {
            const double theta = (A[q + (size_t)ld * q] - A[p + (size_t)ld * p]) / (2. * apq);
            const double t = (theta >= 0 ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
            c = 1. / sqrt(t * t + 1.);
            s = t * c;
          }
        }
        cs[k] = c;
        sn[k] = s;
      }
      __syncthreads();
      // phase B: every (k,k') owns the 2x2 block rows {p,q} x cols {p',q'}:  blk <- J_k' blk J_k'
      const int H = N / 2;
      for (int e = tid; e < H * H; e += kPsdThreads) {
        const int k = e % H, k2 = e / H;
        int p, q, p2, q2;
        rr_pair(r, k, N, p, q);
        rr_pair(r, k2, N, p2, q2);
        const double c = cs[k], s = sn[k], c2 = cs[k2], s2 = sn[k2];
        const bool vq = q < n, vq2 = q2 < n;
        double app = A[p + (size_t)ld * p2];
        double apq = vq2 ? A[p + (size_t)ld * q2] : 0.;
        double aqp = vq ? A[q + (size_t)ld * p2] : 0.;
        double aqq = (vq && vq2) ? A[q + (size_t)ld * q2] : 0.;
        // columns: [a_p' a_q'] <- [c2 a_p' - s2 a_q', s2 a_p' + c2 a_q']
        const double t1 = c2 * app - s2 * apq, t2 = s2 * app + c2 * apq;
        const double t3 = c2 * aqp - s2 * aqq, t4 = s2 * aqp + c2 * aqq;
        // rows: [r_p; r_q] <- [c r_p - s r_q; s r_p + c r_q]
        A[p + (size_t)ld * p2] = c * t1 - s * t3;
        if (vq2) A[p + (size_t)ld * q2] = c * t2 - s * t4;
        if (vq) A[q + (size_t)ld * p2] = s * t1 + c * t3;
        if (vq && vq2) A[q + (size_t)ld * q2] = s * t2 + c * t4;
      }
      // V <- V J (columns p,q of V), rows i coalesced
      for (int e = tid; e < H * n; e += kPsdThreads) {
        const int i = e % n, k = e / n;
        int p, q;
        rr_pair(r, k, N, p, q);
        if (q >= n) continue;
        const double c = cs[k], s = sn[k];
        const double vp = V[i + (size_t)ld * p], vq = V[i + (size_t)ld * q];
        V[i + (size_t)ld * p] = c * vp - s * vq;
        V[i + (size_t)ld * q] = s * vp + c * vq;
      }
      __syncthreads();
    }
  }

  // ---- scale eigenvector columns: W = V diag(sqrt(lambda+)) so X+ = W W' ----
  for (int j = tid; j < ld; j += kPsdThreads) lam[j] = (j < n) ? fmax(A[j + (size_t)ld * j], 0.) : 0.;
  __syncthreads();
  for (int e = tid; e < ld * ld; e += kPsdThreads) {
    const int j = e / ld;
    V[e] *= sqrt(lam[j]);
  }
  __syncthreads();

  // ---- X+ = W W' on the fp64 matrix cores: one wave per 16x16 output tile ----
  // v_mfma_f64_16x16x4_f64: A operand lane l holds A[i=l&15][k=l>>4], B operand holds B[k=l>>4][j=l&15];
  // result reg t of lane l is C[row=(l>>4)+4t][col=l&15].
  const int wave = tid >> 6, lane = tid & 63, nwaves = kPsdThreads / 64;
  const int T = ld / 16;
  for (int tile = wave; tile < T * T; tile += nwaves) {
    const int ti = tile % T, tj = tile / T;
    if (tj > ti) continue;  // lower triangle only
    f64x4 acc = {0., 0., 0., 0.};
    const int li = lane & 15, lk = lane >> 4;
    for (int k0 = 0; k0 < ld; k0 += 4) {
      const double a = V[(ti * 16 + li) + (size_t)ld * (k0 + lk)];  // W[i][k]
      const double b = V[(tj * 16 + li) + (size_t)ld * (k0 + lk)];  // W'[k][j] = W[j][k]
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = ti * 16 + (lane >> 4) + 4 * t, j = tj * 16 + (lane & 15);
      if (i < n && j <= i) {
        const long base = (long)j * n - (long)j * (j - 1) / 2;
        X[base + (i - j)] = (i == j) ? acc[t] : acc[t] * sq2;
      }
    }
  }
}

}  // namespace scship
