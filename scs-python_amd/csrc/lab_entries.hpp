// lab_entries.hpp — PART 2 of include/scs_hip.h: kernel-level entry points for the parity tests and the bench (included inside the extern "C" block of scs_hip.hip)
// (one of the units csrc/scs_hip.hip is assembled from — ONE translation unit, in this order: runtime.hpp, device_csr.hpp, work.hpp
// [+ work_linsys.inl, work_admm.inl, work_residuals.inl, work_solve_ends.inl], io.hpp, setup.hpp, loop.hpp, batch.hpp, the C ABI in scs_hip.hip,
// lab_entries.hpp; split out of the 3 800-line file of rounds 1-5 in round 6 — VERDICT r05 item 6 — without moving a line of code)
#pragma once
// ---- kernel-level entry points (tests / bench) ----
struct TmpStream {
  hipStream_t s = nullptr;
  TmpStream() {
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) throw std::runtime_error("libscs_hip: no HIP device available");
    HIP_CHECK(hipSetDevice(current_device()));
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  }
  ~TmpStream() { if (s) (void)hipStreamDestroy(s); }
};

static void upload_for_spmv(const ScsMatrix *A, int transpose, DeviceCsr &M, hipStream_t s) {
  HostCsr ar;
  csc_to_csr(A->m, A->n, A->p, A->i, A->x, ar);
  const bool host = DeviceCsr::host_setup();
  DeviceCsr T;  // the other orientation: what the device builder of the column-sorted copy reads
  if (transpose) {
    M.upload(A->n, A->m, A->p, A->i, A->x, s);
    if (host) { M.build_cs_host(A->p, A->i, A->x, s, /*kind=*/1); return; }
    if (!cs_wanted(M.rows, M.cols, M.nnz)) return;
    T.upload(A->m, A->n, ar.rowptr.data(), ar.col.data(), ar.val.data(), s, /*allow_slab=*/false);
  } else {
    M.upload(A->m, A->n, ar.rowptr.data(), ar.col.data(), ar.val.data(), s);
    if (host) { M.build_cs_host(ar.rowptr.data(), ar.col.data(), ar.val.data(), s, /*kind=*/0); return; }
    if (!cs_wanted(M.rows, M.cols, M.nnz)) return;
    T.upload(A->n, A->m, A->p, A->i, A->x, s, /*allow_slab=*/false);
  }
  M.build_cs_dev(T, s, /*kind=*/transpose != 0 ? 1 : 0);
}

int scs_hip_spmv(const ScsMatrix *A, const scs_float *x, scs_float *y, int transpose) {
  try {
    set_last_error("");
    refresh_options();
    if (!validate_matrix(A, A->m, A->n)) throw std::runtime_error("invalid matrix");
    TmpStream ts;
    DeviceCsr M;
    upload_for_spmv(A, transpose, M, ts.s);
    DevBuf<double> dx, dy;
    dx.upload(x, M.cols, ts.s);
    dy.upload(y, M.rows, ts.s);
    launch_spmv(M.view(), dx.p, EpiStore{dy.p, 1}, nullptr, ts.s);
    dy.download(y, M.rows, ts.s);
    HIP_CHECK(hipStreamSynchronize(ts.s));
    return 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

static int cs_layout_host_spmv_impl(const ScsMatrix *A, const scs_float *x, scs_float *y, int transpose, int rpt, int split, int piece_len);
int scs_hip_cs_layout_host_spmv(const ScsMatrix *A, const scs_float *x, scs_float *y, int transpose, int rpt, int split) {
  return cs_layout_host_spmv_impl(A, x, y, transpose, rpt, split, 0);
}
// piece_len > 0: the virtual-row layout (spmv_cs.hpp CsView::Rr) — rows longer than max(piece_len, what a count field holds) cut into
// pieces of at most piece_len nonzeros, walked the way the kernels walk it: pass kernel, then one wavefront per long row over its pieces
int scs_hip_cs_layout_host_spmv_pieces(const ScsMatrix *A, const scs_float *x, scs_float *y, int transpose, int piece_len) {
  return cs_layout_host_spmv_impl(A, x, y, transpose, 0, 1, piece_len);
}
static int cs_layout_host_spmv_impl(const ScsMatrix *A, const scs_float *x, scs_float *y, int transpose, int rpt, int split, int piece_len) {
  try {
    set_last_error("");
    refresh_options();
    if (!validate_matrix(A, A->m, A->n)) throw std::runtime_error("invalid matrix");
    HostCsr ar;
    const int *rp = A->p, *ci = A->i;
    const double *v = A->x;
    int rows = A->n, cols = A->m;
    if (!transpose) {
      csc_to_csr(A->m, A->n, A->p, A->i, A->x, ar);
      rp = ar.rowptr.data(); ci = ar.col.data(); v = ar.val.data();
      rows = A->m; cols = A->n;
    }
    HostCs h;
    if (split != 1 && split != 2 && split != 4) throw std::runtime_error("split must be 1, 2 or 4");
    // rows the count fields cannot hold are peeled off the layout, as scs_init does, and summed from the plain CSR
    std::vector<unsigned> mk;
    {
      int R0, rpt0;
      cs_pick_geometry(rows, R0, rpt0, split);
      if (rpt > 0) rpt0 = rpt;
      const int thresh = cs_peel_threshold(rpt0);
      if (opts().cs_peel)
        for (int r = 0; r < rows; ++r)
          if (rp[r + 1] - rp[r] > thresh) {
            if (mk.empty()) mk.assign(((size_t)rows + 31) / 32, 0u);
            mk[r >> 5] |= 1u << (r & 31);
          }
    }
    CsVirtPlan P;
    const bool pieces = piece_len > 0;
    if (pieces) {
      int R0, rpt0;
      cs_pick_geometry(rows, R0, rpt0, 1);
      mk.clear();
      if (!cs_plan_virtual(rp, rows, piece_len, std::max(piece_len, cs_peel_threshold(rpt0)), P)) return 1;
      if (!build_cs_virtual(rp, ci, v, rows, cols, P, h)) return 1;
    } else if (!build_cs(rp, ci, v, rows, cols, h, rpt, split, mk.empty() ? nullptr : mk.data())) return 1;
    std::vector<double> tpart((size_t)P.V, 0.0);
    if (!mk.empty())
      for (int r = 0; r < rows; ++r)
        if (cs_is_peeled(mk.data(), r)) {
          double sacc = 0.;
          for (int q = rp[r]; q < rp[r + 1]; ++q) sacc += v[q] * x[ci[q]];
          y[r] += sacc;
        }
    const int cb = cs_count_bits(h.rpt), mw = cs_meta_words(h.rpt);
    std::vector<double> prod(kCsPass), acc((size_t)kCsThreads * h.rpt), tot((size_t)kCsThreads * h.rpt);
    for (int c = 0; c < h.nchunks; ++c) {
      for (int part = 0; part < h.split; ++part) {  // one workgroup each; split > 1: the partial sums are added in part order
        std::fill(acc.begin(), acc.end(), 0.0);
        const size_t wg = (size_t)c * h.split + part;
        for (int g = h.passptr[wg]; g < h.passptr[wg + 1]; ++g) {
          const int2 pi = h.pinfo[g];
          const size_t o = (size_t)g * kCsPass;
          for (int sp = 0; sp < kCsPass; ++sp) {  // the whole pass, padding included, as the kernel does
            const unsigned id = h.idx[o + sp];
            prod[id & (kCsPass - 1)] = h.val[o + sp] * x[pi.x + (int)(id >> kCsSlotBits)];
          }
          for (int t = 0; t < kCsThreads; ++t) {
            const unsigned long long mw0 = h.meta[((size_t)g * kCsThreads + t) * mw];
            int off = (int)(mw0 & 0xffff);
            unsigned long long w = mw0 >> 16;
            for (int j = 0; j < h.rpt; ++j) {
              if (h.rpt == 16 && j == 8) w = h.meta[((size_t)g * kCsThreads + t) * mw + 1];
              const int n = (int)(w & ((1ull << cb) - 1));
              w >>= cb;
              double sacc = acc[(size_t)j * kCsThreads + t];
              for (int k = 0; k < n; ++k) sacc += prod[off + k];
              acc[(size_t)j * kCsThreads + t] = sacc;
              off += n;
            }
          }
        }
        if (part == 0) tot = acc;
        else for (size_t i = 0; i < tot.size(); ++i) tot[i] += acc[i];
      }
      const int Rr = pieces ? P.Rr : h.R;
      for (int rl = 0; rl < h.R; ++rl) {
        if (rl < Rr) {
          const long r = (long)c * Rr + rl;
          if (r < rows && !(pieces && cs_is_peeled(P.mask.data(), (int)r))) y[r] += tot[rl];
        } else {
          const long p = (long)c * P.Rp + (rl - Rr);
          if (p < P.V) tpart[(size_t)p] = tot[rl];
        }
      }
    }
    for (const int4 &b : P.blk) {  // k_spmv_peeled<Epi, PIECES>: lanes stride over the row's pieces, then the wave's shuffle tree
      double lane[64];
      for (int l = 0; l < 64; ++l) {
        double a = 0.;
        for (int k = b.z + l; k < b.w; k += 64) a += tpart[(size_t)k];
        lane[l] = a;
      }
      for (int o = 32; o > 0; o >>= 1) {
        double nxt[64];
        for (int l = 0; l < 64; ++l) nxt[l] = lane[l] + (l + o < 64 ? lane[l + o] : lane[l]);
        std::memcpy(lane, nxt, sizeof(lane));
      }
      y[b.x] += lane[0];
    }
    return 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

double scs_hip_spmv_bench(const ScsMatrix *A, int transpose, int reps) {
  try {
    set_last_error("");
    refresh_options();
    TmpStream ts;
    DeviceCsr M;
    upload_for_spmv(A, transpose, M, ts.s);
    std::vector<double> hx(M.cols);
    for (int i = 0; i < M.cols; ++i) hx[i] = 1.0 + 1e-3 * (i % 977);
    DevBuf<double> dx, dy;
    dx.upload(hx.data(), M.cols, ts.s);
    dy.alloc_zero(M.rows, ts.s);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_spmv(M.view(), dx.p, EpiStore{dy.p, 0}, nullptr, ts.s);
    HIP_CHECK(hipEventRecord(e0, ts.s));
    for (int i = 0; i < reps; ++i) launch_spmv(M.view(), dx.p, EpiStore{dy.p, 0}, nullptr, ts.s);
    HIP_CHECK(hipEventRecord(e1, ts.s));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)ms / reps;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1.0;
  }
}

// One-shot projections (tests, generators): a temporary workspace on the CURRENT device.  The multi-CU PSD sweep kernel may run here
// (fl is allocated), so its error flag is read behind every synchronisation: a barrier that timed out — another process holds part
// of the GPU — opened every barrier of the launch and left garbage; the call is then repeated from its inputs with one workgroup per
// matrix, the path a solve takes after a SpinTimeout.
static void oneshot_cone_work(ScsHipWork &w, const ScsCone *k, scs_int m, int warm, hipStream_t s, bool no_spin) {
  if (!build_cone(k, w.cone) || w.cone.m != m) throw std::runtime_error("invalid cone");
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  w.device = dev;  // (spin_chain(), the occupancy query of psd_mc_members)
  w.stream = s;
  w.owns_stream = false;
  w.m = m;
  w.psd_warm = warm;
  if (no_spin) w.psd_mc_cap = 0;
  upload_cone_meta(&w);
  w.fl.alloc_zero(F_COUNT, s);  // (k_psd_sweep_mc polls its error flag while it waits at a barrier)
  w.sc.alloc_zero(S_COUNT, s);
  const double one = 1.0;
  HIP_CHECK(hipMemcpyAsync(w.sc.p + S_BOX_T, &one, sizeof(double), hipMemcpyHostToDevice, s));
}
static bool oneshot_spin_error(ScsHipWork &w, hipStream_t s) {  // the stream is idle
  int err = 0;
  HIP_CHECK(hipMemcpyAsync(&err, w.fl.p + F_PERSIST_ERR, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  return err != 0;
}

int scs_hip_proj_cone(scs_float *x, const ScsCone *k, scs_int m, int dual) {
  try {
    set_last_error("");
    refresh_options();
    for (int attempt = 0; attempt < 2; ++attempt) {
      ScsHipWork w;
      TmpStream ts;
      oneshot_cone_work(w, k, m, /*warm=*/0, ts.s, attempt > 0);
      DevBuf<double> dx;
      dx.upload(x, m, ts.s);
      if (w.cone.z + w.cone.l > 0)
        hipLaunchKernelGGL(k_proj_zl, dim3(ceil_div(w.cone.z + w.cone.l, kConeThreads)), dim3(kConeThreads), 0, ts.s, dx.p,
                           w.cone.z, w.cone.l, dual);
      w.project_nonlinear_cones(dx.p, dual);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(ts.s));
      if (oneshot_spin_error(w, ts.s)) {
        if (attempt > 0) throw std::runtime_error("cone projection: a device-side barrier timed out");
        g_spin_fallbacks.fetch_add(1);
        continue;  // x is untouched: again, without the spinning kernel
      }
      dx.download(x, m, ts.s);
      HIP_CHECK(hipStreamSynchronize(ts.s));
      return 0;
    }
    return -1;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

int scs_hip_proj_cone_seq(scs_float *xs, const ScsCone *k, scs_int m, int dual, int count, scs_float *stats, int stats_cap) {
  try {
    set_last_error("");
    refresh_options();
    if (count < 0 || !xs) throw std::runtime_error("invalid sequence");
    std::vector<double> out((size_t)count * (size_t)std::max(m, 0));  // the inputs stay intact until the whole sequence went through
    for (int attempt = 0; attempt < 2; ++attempt) {
      ScsHipWork w;
      TmpStream ts;
      // warm = 1: as inside the ADMM loop, the eigenvectors (and every other cone's warm-start state) carry over from call to call
      oneshot_cone_work(w, k, m, /*warm=*/1, ts.s, attempt > 0);
      DevBuf<double> dx;
      dx.alloc((size_t)std::max(m, 1));
      bool spin_err = false;
      for (int c = 0; c < count && !spin_err; ++c) {
        HIP_CHECK(hipMemcpyAsync(dx.p, xs + (size_t)c * m, sizeof(double) * m, hipMemcpyHostToDevice, ts.s));
        if (w.cone.z + w.cone.l > 0)
          hipLaunchKernelGGL(k_proj_zl, dim3(ceil_div(w.cone.z + w.cone.l, kConeThreads)), dim3(kConeThreads), 0, ts.s, dx.p,
                             w.cone.z, w.cone.l, dual);
        w.project_nonlinear_cones(dx.p, dual);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out.data() + (size_t)c * m, dx.p, sizeof(double) * m, hipMemcpyDeviceToHost, ts.s));
        HIP_CHECK(hipStreamSynchronize(ts.s));
        spin_err = oneshot_spin_error(w, ts.s);
      }
      if (spin_err) {  // the warm-start state behind the failed projection is garbage: the whole sequence again, without the spinning kernel
        if (attempt > 0) throw std::runtime_error("cone projection: a device-side barrier timed out");
        g_spin_fallbacks.fetch_add(1);
        continue;
      }
      int nst = 0;
      if (stats && stats_cap > 0) {
        nst = std::min(stats_cap, w.n_psd_big);
        for (int c = 0; c < nst; ++c) {
          double st[kPsdStateDoubles];
          const long at = w.psd_woff_h[(size_t)c] + psd_scratch_doubles(w.psd_order_h[(size_t)c]) - kPsdStateDoubles;
          HIP_CHECK(hipMemcpy(st, w.psd_scratch.p + at, sizeof st, hipMemcpyDeviceToHost));
          stats[8 * c + 0] = st[9]; stats[8 * c + 1] = st[10]; stats[8 * c + 2] = st[8]; stats[8 * c + 3] = st[11]; stats[8 * c + 4] = st[7];
          stats[8 * c + 5] = st[12]; stats[8 * c + 6] = st[13]; stats[8 * c + 7] = st[14];
        }
      }
      if (!out.empty()) std::memcpy(xs, out.data(), out.size() * sizeof(double));
      return nst;
    }
    return -1;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

static int kkt_solve_entry(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, scs_float *rhs, scs_float tol,
                           scs_int *cg_iters, bool dense);
int scs_hip_kkt_solve(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, scs_float *rhs, scs_float tol,
                      scs_int *cg_iters) {
  return kkt_solve_entry(A, P, diag_r, rhs, tol, cg_iters, false);
}
int scs_hip_kkt_solve_dense(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, scs_float *rhs) {
  return kkt_solve_entry(A, P, diag_r, rhs, 0., nullptr, true);
}
static int kkt_solve_entry(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, scs_float *rhs, scs_float tol,
                           scs_int *cg_iters, bool dense) {
  try {
    set_last_error("");
    refresh_options();
    if (!validate_matrix(A, A->m, A->n)) throw std::runtime_error("invalid A");
    ScsHipWork w;
    TmpStream ts;
    hipStream_t s = ts.s;
    w.stream = s;
    w.owns_stream = false;
    const int n = A->n, m = A->m;
    w.n = n; w.m = m; w.l = (long)n + m + 1;
    w.has_P = P != nullptr;
    HIP_CHECK(hipHostMalloc((void **)&w.h_flags, sizeof(int) * F_COUNT));
    w.At.upload(n, m, A->p, A->i, A->x, s);
    {
      HostCsr ar;
      csc_to_csr(m, n, A->p, A->i, A->x, ar);
      w.Ar.upload(m, n, ar.rowptr.data(), ar.col.data(), ar.val.data(), s);
    }
    if (P) {
      HostCsr pf;
      std::vector<double> pdiag;
      sym_expand(n, P->p, P->i, P->x, pf, pdiag);
      w.Pf.upload(n, n, pf.rowptr.data(), pf.col.data(), pf.val.data(), s);
      w.Pdiag.upload(pdiag.data(), n, s);
    }
    if (!DeviceCsr::host_setup()) {
      w.At.build_cs_dev(w.Ar, s, /*kind=*/1);
      w.Ar.build_cs_dev(w.At, s, /*kind=*/0);
      if (P) w.Pf.build_cs_dev(w.Pf, s, /*kind=*/2);
    }
    std::vector<double> dr(w.l, 10.0);
    std::copy(diag_r, diag_r + n + m, dr.begin());
    w.diag_r.upload(dr.data(), w.l, s);
    for (DevBuf<double> *b : {&w.cg_b, &w.cg_p, &w.cg_r, &w.cg_Gp, &w.cg_M, &w.ws}) b->alloc_zero(n, s);
    w.cg_ticket.alloc_zero(1, s);
    w.tmp_m.alloc_zero(m, s);
    w.part.alloc_zero((size_t)std::max({w.At.nblk, w.Ar.nblk, w.At.nwg(), w.Ar.nwg(), kMaxVecBlocks}) * kMaxEpiReductions, s);
    w.part2.alloc_zero(2 * kMaxVecBlocks, s);
    w.sc.alloc_zero(S_COUNT, s);
    w.fl.alloc_zero(F_COUNT, s);
    if (dense) {
      if (n > kDenseMaxN) throw std::runtime_error("dense KKT solve: n too large");
      w.linsys = 1;
      w.dense_alloc();
      w.dense_refactor();
    } else {
      hipLaunchKernelGGL(k_precond, dim3(vec_blocks(n)), dim3(kVecThreads), 0, s, w.At.rowptr.p, w.At.col.p, w.At.val.p,
                         w.diag_r.p, P ? w.Pdiag.p : (const double *)nullptr, w.cg_M.p, n);
    }
    DevBuf<double> drhs;
    drhs.upload(rhs, (size_t)n + m, s);
    const int its = w.kkt_solve(drhs.p, tol);
    drhs.download(rhs, (size_t)n + m, s);
    HIP_CHECK(hipStreamSynchronize(s));
    if (cg_iters) *cg_iters = its;
    return 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

int scs_hip_normalize(ScsMatrix *A, ScsMatrix *P, scs_float *b, scs_float *c, const ScsCone *k, scs_float *D, scs_float *E,
                      scs_float *sigma) {
  try {
    set_last_error("");
    refresh_options();
    HostCone cone;
    if (!build_cone(k, cone) || cone.m != A->m) throw std::runtime_error("invalid cone");
    if (!validate_matrix(A, A->m, A->n)) throw std::runtime_error("invalid A");
    TmpStream ts;
    hipStream_t s = ts.s;
    const int m = A->m, n = A->n;
    DeviceCsr At, Ar, Pf;
    At.upload(n, m, A->p, A->i, A->x, s, false);
    {
      HostCsr ar;
      csc_to_csr(m, n, A->p, A->i, A->x, ar);
      Ar.upload(m, n, ar.rowptr.data(), ar.col.data(), ar.val.data(), s, false);
    }
    if (P) {
      HostCsr pf;
      std::vector<double> pdiag;
      sym_expand(n, P->p, P->i, P->x, pf, pdiag);
      Pf.upload(n, n, pf.rowptr.data(), pf.col.data(), pf.val.data(), s, false);
    }
    DevBuf<double> dD, dE;
    device_normalize(At, Ar, P ? &Pf : nullptr, cone, dD, dE, s);
    HostScaling sc;
    sc.D.resize(m);
    sc.E.resize(n);
    dD.download(sc.D.data(), m, s);
    dE.download(sc.E.data(), n, s);
    At.val.download(A->x, (size_t)A->p[n], s);  // CSR(A') order == the caller's CSC order
    HIP_CHECK(hipStreamSynchronize(s));
    if (P)
      for (int j = 0; j < n; ++j)
        for (int q = P->p[j]; q < P->p[j + 1]; ++q) P->x[q] *= sc.E[P->i[q]] * sc.E[j];
    normalize_b_c(sc, b, m, c, n);
    std::copy(sc.D.begin(), sc.D.end(), D);
    std::copy(sc.E.begin(), sc.E.end(), E);
    *sigma = sc.sigma;
    return 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

// ---- Anderson acceleration as a standalone object (row a6): the interface of scs_source/src/aa.c
// (aa_init / aa_apply / aa_safeguard / aa_reset / aa_finish, named at R:meson.build:187) on host vectors — tests
// drive it step by step next to the CPU checker.  Inside scs_solve the same DeviceAa works on the resident iterate.
struct ScsHipAa {
  int device = 0;
  hipStream_t stream = nullptr;
  DeviceAa aa;
  DevBuf<double> f, x;
  DevBuf<int> bad;
  ~ScsHipAa() { if (stream) (void)hipStreamDestroy(stream); }
};

ScsHipAa *scs_hip_aa_init(scs_int dim, scs_int mem, scs_int type1, scs_float regularization, scs_float relaxation,
                          scs_float safeguard_factor, scs_float max_weight_norm) {
  try {
    set_last_error("");
    refresh_options();
    if (dim <= 0 || mem < 0) throw std::runtime_error("invalid AA dimensions");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) throw std::runtime_error("libscs_hip: no HIP device available");
    std::unique_ptr<ScsHipAa> a(new ScsHipAa());
    a->device = current_device();
    HIP_CHECK(hipSetDevice(a->device));
    HIP_CHECK(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    a->aa.init(dim, mem, type1, regularization, relaxation, safeguard_factor, max_weight_norm, a->stream);
    a->f.alloc_zero((size_t)dim, a->stream);
    a->x.alloc_zero((size_t)dim, a->stream);
    a->bad.alloc_zero(1, a->stream);
    HIP_CHECK(hipStreamSynchronize(a->stream));
    return a.release();
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return nullptr;
  }
}

scs_float scs_hip_aa_apply(ScsHipAa *a, scs_float *f, const scs_float *x) {
  if (!a || !f || !x) return NAN;
  try {
    set_last_error("");
    refresh_options();
    HIP_CHECK(hipSetDevice(a->device));
    a->f.upload(f, (size_t)a->aa.dim, a->stream);
    a->x.upload(x, (size_t)a->aa.dim, a->stream);
    const double nrm = a->aa.apply(a->f.p, a->x.p);
    a->f.download(f, (size_t)a->aa.dim, a->stream);
    HIP_CHECK(hipStreamSynchronize(a->stream));
    return nrm;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return NAN;
  }
}

scs_int scs_hip_aa_safeguard(ScsHipAa *a, scs_float *f_new, scs_float *x_new) {
  if (!a || !f_new || !x_new) return -2;
  try {
    set_last_error("");
    refresh_options();
    HIP_CHECK(hipSetDevice(a->device));
    if (!a->aa.success) return 0;  // nothing to test (and no asynchronous upload of the caller's buffers left in flight)
    a->f.upload(f_new, (size_t)a->aa.dim, a->stream);
    a->x.upload(x_new, (size_t)a->aa.dim, a->stream);
    a->aa.safeguard(a->f.p, a->x.p, a->bad.p);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, a->bad.p, sizeof(int), hipMemcpyDeviceToHost, a->stream));
    a->f.download(f_new, (size_t)a->aa.dim, a->stream);
    a->x.download(x_new, (size_t)a->aa.dim, a->stream);
    HIP_CHECK(hipStreamSynchronize(a->stream));
    a->aa.safeguard_verdict(bad != 0);
    return bad ? -1 : 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -2;
  }
}

void scs_hip_aa_reset(ScsHipAa *a) {
  if (a) a->aa.reset();
}
void scs_hip_aa_get_stats(const ScsHipAa *a, ScsAaStats *st) {
  if (a && st) *st = a->aa.st;
}
scs_int scs_hip_aa_last_gamma(const ScsHipAa *a, scs_float *gamma) {
  if (!a) return 0;
  if (gamma) std::copy(a->aa.last_gamma.begin(), a->aa.last_gamma.end(), gamma);
  return (scs_int)a->aa.last_gamma.size();
}
void scs_hip_aa_finish(ScsHipAa *a) {
  if (!a) return;
  (void)hipSetDevice(a->device);
  if (a->stream) (void)hipStreamSynchronize(a->stream);
  delete a;
}

__global__ void k_copy4(const double4 *__restrict__ src, double4 *dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

double scs_hip_copy_bandwidth(size_t bytes, int reps) {
  try {
    set_last_error("");
    refresh_options();
    TmpStream ts;
    const size_t n4 = bytes / sizeof(double4);
    DevBuf<double4> a, b;
    a.alloc_zero(n4, ts.s);
    b.alloc_zero(n4, ts.s);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_copy4, dim3(4096), dim3(256), 0, ts.s, a.p, b.p, n4);
    HIP_CHECK(hipEventRecord(e0, ts.s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_copy4, dim3(4096), dim3(256), 0, ts.s, a.p, b.p, n4);
    HIP_CHECK(hipEventRecord(e1, ts.s));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 2.0 * (double)(n4 * sizeof(double4)) * reps / (ms * 1e-3) / 1e9;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1.0;
  }
}
