// setup.hpp — scs_init: cone metadata upload and init_impl (validation, uploads, layouts, equilibration, vectors, AA workspace, R / preconditioner / g)
// (one of the units csrc/scs_hip.hip is assembled from — ONE translation unit, in this order: runtime.hpp, device_csr.hpp, work.hpp
// [+ work_linsys.inl, work_admm.inl, work_residuals.inl, work_solve_ends.inl], io.hpp, setup.hpp, loop.hpp, batch.hpp, the C ABI in scs_hip.hip,
// lab_entries.hpp; split out of the 3 800-line file of rounds 1-5 in round 6 — VERDICT r05 item 6 — without moving a line of code)
#pragma once
// ================================================================ init
static void upload_cone_meta(ScsHipWork *w) {
  hipStream_t s = w->stream;
  const HostCone &c = w->cone;
  std::vector<int> off, dim, big;
  int o = c.off_q, max_small = 0;
  for (size_t i = 0; i < c.q.size(); ++i) {
    off.push_back(o);
    dim.push_back(c.q[i]);
    if (c.q[i] > kSocBig) big.push_back((int)i);
    else max_small = std::max(max_small, (int)c.q[i]);
    o += c.q[i];
  }
  w->n_soc = (int)off.size();
  w->soc_G = soc_group(max_small);
  w->n_soc_big = (int)big.size();
  if (w->n_soc) { w->soc_off.upload(off.data(), off.size(), s); w->soc_dim.upload(dim.data(), dim.size(), s); }
  if (w->n_soc_big) w->soc_big.upload(big.data(), big.size(), s);
  if (!c.p.empty()) w->pow_a.upload(c.p.data(), c.p.size(), s);
  if (c.bsize > 1) {
    w->box_bl.upload(c.bl.data(), c.bl.size(), s);
    w->box_bu.upload(c.bu.data(), c.bu.size(), s);
    if (!w->box_bl_orig.p) {  // (scs_init uploaded the originals before the row scaling; the standalone entry points have none)
      w->box_bl_orig.upload(c.bl.data(), c.bl.size(), s);
      w->box_bu_orig.upload(c.bu.data(), c.bu.size(), s);
    }
  }
  std::vector<int> poff, pord;
  std::vector<long> woff;
  long wtot = 0;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0: orders > kPsdSmallMax (block kernel), pass 1: the one-wave kernel's
    o = c.off_s;
    for (int sdim : c.s) {
      if ((sdim > kPsdSmallMax) == (pass == 0)) {
        poff.push_back(o);
        pord.push_back(sdim);
        woff.push_back(wtot);
        wtot += psd_scratch_doubles(sdim);
      }
      o += (int)sd_size(sdim);
    }
    if (pass == 0) w->n_psd_big = (int)poff.size();
  }
  w->n_psd = (int)poff.size();
  if (w->n_psd) {
    w->psd_off.upload(poff.data(), poff.size(), s);
    w->psd_order.upload(pord.data(), pord.size(), s);
    w->psd_woff.upload(woff.data(), woff.size(), s);
    w->psd_woff_h = woff;
    w->psd_order_h = pord;
  }
  std::vector<int> coff, cord, cpoff, cpord;
  std::vector<long> csoff, cwoff;
  long stot = 0;
  for (int pass = 0; pass < 2; ++pass) {
    o = c.off_cs;
    for (int k : c.cs) {
      if ((2 * k > kPsdSmallMax) == (pass == 0)) {
        coff.push_back(o);
        cord.push_back(k);
        csoff.push_back(stot);
        cpoff.push_back((int)stot);
        cpord.push_back(2 * k);
        cwoff.push_back(wtot);
        wtot += psd_scratch_doubles(2 * k);
        stot += sd_size(2 * k);
      }
      o += k * k;
    }
    if (pass == 0) w->n_cs_big = (int)coff.size();
  }
  w->n_cs = (int)coff.size();
  if (w->n_cs) {
    w->cs_off.upload(coff.data(), coff.size(), s);
    w->cs_order.upload(cord.data(), cord.size(), s);
    w->cs_soff.upload(csoff.data(), csoff.size(), s);
    w->cs_poff.upload(cpoff.data(), cpoff.size(), s);
    w->cs_porder.upload(cpord.data(), cpord.size(), s);
    w->cs_woff.upload(cwoff.data(), cwoff.size(), s);
    w->cs_stage.alloc_zero((size_t)std::max(stot, 1L), s);
  }
  if (w->n_psd || w->n_cs) w->psd_scratch.alloc_zero((size_t)std::max(wtot, 1L), s);
  {
    int big_total = 0, max_order = 0;
    for (int sdim : c.s)
      if (sdim > kPsdSmallMax) { ++big_total; max_order = std::max(max_order, sdim); }
    for (int k : c.cs)
      if (2 * k > kPsdSmallMax) { ++big_total; max_order = std::max(max_order, 2 * k); }
    w->psd_max_np = (int)psd_np(std::max(max_order, 2));
    w->psd_max_tiles = w->psd_max_np / 16;
    // split mode: one CU per matrix would leave at least half of the GPU idle.  Its V update keeps a 16-row strip of V in LDS
    // (16 x NP doubles): orders above 1280 do not fit and take the one-workgroup-per-matrix kernel (any order up to 16 kPsdMaxH)
    w->psd_split = big_total > 0 && big_total <= 128 && (size_t)16 * w->psd_max_np * sizeof(double) <= 160 * 1024;
    if (opts().psd_split >= 0) w->psd_split = big_total > 0 && opts().psd_split == 1;  // SCS_HIP_PSD_SPLIT: A/B and tests
  }
  HIP_CHECK(hipStreamSynchronize(s));
}

static ScsHipWork *init_impl(const ScsData *d, const ScsCone *k, const ScsSettings *stgs, int linsys = 0) {
  const double t0 = now_ms();
  refresh_options();  // the environment as it is NOW: this workspace keeps what it is created with (options.hpp)
  if (linsys == 0) linsys = opts().linsys_dense ? 2 : 1;
  if (linsys != 1 && linsys != 2) throw std::runtime_error("unknown linear-system solver kind");
  if (!d || !k || !stgs) throw std::runtime_error("null argument");
  if (d->m <= 0 || d->n <= 0 || !d->A || !d->b || !d->c) throw std::runtime_error("invalid data dimensions");
  if (!validate_matrix(d->A, d->m, d->n)) throw std::runtime_error("invalid A matrix");
  if (d->P && !validate_matrix(d->P, d->n, d->n)) throw std::runtime_error("invalid P matrix");
  if (!(stgs->max_iters > 0) || !(stgs->eps_abs >= 0) || !(stgs->eps_rel >= 0) || !(stgs->eps_infeas >= 0) ||
      !(stgs->alpha > 0 && stgs->alpha < 2) || !(stgs->rho_x > 0) || !(stgs->scale > 0) ||
      !(stgs->acceleration_interval > 0) || stgs->acceleration_lookback < 0 ||
      !(stgs->time_limit_secs >= 0))
    throw std::runtime_error("invalid settings");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    throw std::runtime_error("libscs_hip: no HIP device available (this backend has no CPU fallback)");
  HIP_CHECK(hipSetDevice(current_device()));

  std::unique_ptr<ScsHipWork> w(new ScsHipWork());
  w->device = current_device();
  if (!build_cone(k, w->cone)) throw std::runtime_error("invalid cone");
  if (w->cone.m != d->m) throw std::runtime_error("cone dimensions do not match m");
  const int n = d->n, m = d->m;
  if (linsys == 2 && n > kDenseMaxN)
    throw std::runtime_error("hip_dense: n = " + std::to_string(n) + " exceeds " + std::to_string(kDenseMaxN) +
                             " (the explicit inverse of the reduced KKT matrix would need " + std::to_string((long)n * n * 8 / 1000000) +
                             " MB); use the indirect solver");
  w->linsys = linsys == 2 ? 1 : 0;
  w->n = n; w->m = m; w->l = (long)n + m + 1;
  w->stgs = *stgs;
  if (stgs->write_data_filename) w->write_data_filename = stgs->write_data_filename;
  if (stgs->log_csv_filename) w->log_csv_filename = stgs->log_csv_filename;
  w->stgs.write_data_filename = nullptr;
  w->stgs.log_csv_filename = nullptr;
  w->scale = stgs->scale;
  w->has_P = d->P != nullptr;
  if (!w->write_data_filename.empty()) write_problem_data(w->write_data_filename.c_str(), d, k, stgs);
  w->b_orig.assign(d->b, d->b + m);
  w->c_orig.assign(d->c, d->c + n);
  for (double x : w->b_orig) w->nm_b_orig = std::max(w->nm_b_orig, std::fabs(x));
  for (double x : w->c_orig) w->nm_c_orig = std::max(w->nm_c_orig, std::fabs(x));

  {
#ifdef SCS_HIP_LABS
    w->graphs_enabled = opts().graph;   // hipGraph replay of the iteration when the host looks at every iteration
#endif
    w->pipelined = opts().pipeline;     // SCS_HIP_PIPELINE=0: the host looks at the CG flags in every iteration
    w->pipe_chunk_override = opts().pipe_chunk;
  }
  if (!w->pipelined && w->graphs_enabled) {  // hipGraph capture needs a stream nobody else enqueues on: a private one
    HIP_CHECK(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
  } else {
    w->stream = g_streams.acquire(w->device, &w->stream_shared);
    w->pooled_stream = true;
  }
  for (auto &e : w->ev) HIP_CHECK(hipEventCreate(&e));
  {  // one pinned, device-mapped block: [h_pin 256 f64 | AA h_pin 256 f64 | params 2 x P_COUNT f64 | flags 3 x F_COUNT i32]
    char *blk = (char *)g_pinned.acquire();
    w->pinned_block = blk;
    std::memset(blk, 0, kPinnedBlockBytes);
    w->h_pin = (double *)blk;
    w->aa.h_pin = (double *)blk + 256;
    w->aa.owns_pin = false;
    w->h_params_base = (double *)blk + 512;
    w->h_flags = (int *)((double *)blk + 512 + 2 * P_COUNT);
    w->h_flags_slot[0] = w->h_flags + F_COUNT;
    w->h_flags_slot[1] = w->h_flags + 2 * F_COUNT;
    static_assert((512 + 2 * P_COUNT) * sizeof(double) + 3 * F_COUNT * sizeof(int) <= kPinnedBlockBytes, "pinned block too small");
    HIP_CHECK(hipHostGetDevicePointer((void **)&w->d_params_base, w->h_params_base, 0));
  }
  w->h_params = w->h_params_base;
  w->d_params = w->d_params_base;
  for (int i = 0; i < 2; ++i) HIP_CHECK(hipEventCreateWithFlags(&w->ev_iter[i], hipEventDisableTiming));
  hipStream_t s = w->stream;
  // small problems (config 5: a batch of them) take their device memory from one arena (common.hpp) instead of ~100
  // separate allocations; SCS_HIP_ARENA=0 restores exact allocations (A/B)
  {
    const long annz = d->A->p[n];
    if (opts().arena && annz <= (1L << 18) && w->l <= (1L << 17)) {
      w->arena.reset(new Arena());
      w->arena->stream = s;
      {  // ~40 doubles per row / column of vectors + 3 matrix layouts of 12 B per nonzero + the Anderson history, rounded up to a power of two
        const long mem = std::max(0, stgs->acceleration_lookback);
        size_t est = (size_t)(8 * (40 + 3 * mem) * w->l + 3 * 12 * annz + (256 << 10));
        size_t c = 256 << 10;
        while (c < est && c < Arena::kChunkBytes) c <<= 1;
        w->arena->first_chunk = c;
      }
    }
  }
  ArenaScope arena_scope(w->arena.get());
  const bool setup_timing = (opts().debug & DBG_SETUP) != 0;  // SCS_HIP_DEBUG=setup: where does scs_init spend its time
  double t_mark = now_ms();
  auto mark = [&](const char *what) {
    if (!setup_timing) return;
    HIP_CHECK(hipStreamSynchronize(s));
    const double t = now_ms();
    std::fprintf(stderr, "[scs-hip setup] %-34s %8.1f ms\n", what, t - t_mark);
    t_mark = t;
  };
  mark("validation, cone, host copies");

  // ---- matrices to HBM (raw): CSC(A) as CSR(A'), explicit CSR(A), full CSR(P) ----
  w->normalized = stgs->normalize != 0;
  // Device path (default): upload the caller's CSC once, transpose and (after the equilibration) build the
  // L2-blocked copies on the device; the host builders remain for SCS_HIP_SETUP=host and for rows too long to sort.
  const bool host_build = DeviceCsr::host_setup();
  bool slabs_pending = false;
  w->At.upload(n, m, d->A->p, d->A->i, d->A->x, s, /*allow_slab=*/host_build);
  mark("A' upload (+ slab build on the host)");
  HostCsr ar, pf;  // host copies of the index arrays: only filled on the host paths
  if (host_build || !w->Ar.transpose_from(w->At, s)) {
    csc_to_csr(m, n, d->A->p, d->A->i, d->A->x, ar);
    mark("CSC -> CSR on the host");
    w->Ar.upload(m, n, ar.rowptr.data(), ar.col.data(), ar.val.data(), s, /*allow_slab=*/host_build);
    mark("A upload (+ slab build on the host)");
    slabs_pending = !host_build;
  } else {
    mark("CSC -> CSR on the device");
    slabs_pending = true;
  }
  if (w->has_P) {
    std::vector<double> pdiag;
    sym_expand(n, d->P->p, d->P->i, d->P->x, pf, pdiag);
    w->Pf.upload(n, n, pf.rowptr.data(), pf.col.data(), pf.val.data(), s, /*allow_slab=*/host_build);
    w->px.alloc_zero(n, s);
  }
  if (w->cone.bsize > 1) {  // the caller's box bounds, before the row scaling touches the working copies
    w->box_bl_orig.upload(w->cone.bl.data(), w->cone.bl.size(), s);
    w->box_bu_orig.upload(w->cone.bu.data(), w->cone.bu.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
  // ---- K12: equilibrate on the device, in place in all resident layouts ----
  if (w->normalized) {
    device_normalize(w->At, w->Ar, w->has_P ? &w->Pf : nullptr, w->cone, w->D, w->E, s);
    w->scal.D.resize(m);
    w->scal.E.resize(n);
    w->D.download(w->scal.D.data(), m, s);
    w->E.download(w->scal.E.data(), n, s);
    HIP_CHECK(hipStreamSynchronize(s));
    if (w->cone.bsize > 1) {  // box bounds follow the row scaling: bl_j <- bl_j D_{j+1} / D_0
      const double *Db = &w->scal.D[w->cone.off_box];
      for (int j = 0; j < w->cone.bsize - 1; ++j) {
        w->cone.bu[j] = (w->cone.bu[j] >= 1e15) ? INFINITY : Db[j + 1] * w->cone.bu[j] / Db[0];
        w->cone.bl[j] = (w->cone.bl[j] <= -1e15) ? -INFINITY : Db[j + 1] * w->cone.bl[j] / Db[0];
      }
    }
  }
  mark("equilibration (device)");
  for (DeviceCsr *M : {&w->At, &w->Ar, &w->Pf}) M->refresh_slab(s, true);
  if (slabs_pending) {
    // large matrices: column-sorted pass copy (spmv_cs.hpp), each built from the other orientation's CSR;
    // the L2-blocked slab copy only where the pattern does not fit that format
    // A' products feed the CG update, which takes Gp as the sum of two partial vectors: two workgroups per chunk
    if (!w->At.build_cs_dev(w->Ar, s, /*kind=*/1)) w->At.build_slab_dev(s);
    if (!w->Ar.build_cs_dev(w->At, s, /*kind=*/0)) w->Ar.build_slab_dev(s);
    if (w->has_P && !w->Pf.build_cs_dev(w->Pf, s, /*kind=*/2)) w->Pf.build_slab_dev(s);
  }
  if (host_build) {  // SCS_HIP_SETUP=host: the column-sorted copies from the host builder, on the equilibrated values
    std::vector<double> hv;
    auto host_cs = [&](DeviceCsr &M, const int *rp, const int *ci, int kind) {
      if (!cs_wanted(M.rows, M.cols, M.nnz)) return;
      hv.resize((size_t)M.nnz);
      M.val.download(hv.data(), (size_t)M.nnz, s);
      HIP_CHECK(hipStreamSynchronize(s));
      M.build_cs_host(rp, ci, hv.data(), s, kind);
    };
    host_cs(w->At, d->A->p, d->A->i, 1);
    host_cs(w->Ar, ar.rowptr.data(), ar.col.data(), 0);
    if (w->has_P) host_cs(w->Pf, pf.rowptr.data(), pf.col.data(), 2);
  }
  mark("column-sorted / L2-blocked copies, value refresh");
  if (w->has_P) {  // diagonal of the (scaled) P for the Jacobi preconditioner
    w->Pdiag.alloc_zero(n, s);
    hipLaunchKernelGGL(k_csr_diag, dim3(vec_blocks(n)), dim3(kVecThreads), 0, s, w->Pf.rowptr.p, w->Pf.col.p, w->Pf.val.p, n,
                       w->Pdiag.p);
  }
  // ---- vectors ----
  const long l = w->l;
  for (DevBuf<double> *b : {&w->v, &w->v_prev, &w->u, &w->ut, &w->rsk, &w->diag_r}) b->alloc_zero(l, s);
  w->g.alloc_zero(l, s);
  w->h.alloc_zero(l, s);
  for (DevBuf<double> *b : {&w->cg_b, &w->cg_p, &w->cg_r, &w->cg_Gp, &w->cg_M, &w->ws}) b->alloc_zero(n, s);
  w->cg_ticket.alloc_zero(1, s);
  w->tmp_m.alloc_zero(m, s);
  w->ensure_solution_mirror();
  w->solx.alloc_zero(n, s);
  w->soly.alloc_zero(m, s);
  w->sols.alloc_zero(m, s);
  // (x 8 until round 3: the residual epilogues leave 9 and 10 values per workgroup — with more than 1638 workgroups, i.e. the CSR-stream
  // layout of a matrix beyond ~3.4 M nonzeros, their partials ran past the buffer: a memory fault at 9419 row blocks)
  w->part_len = std::max({w->At.nblk, w->Ar.nblk, w->At.nwg(), w->Ar.nwg(), w->has_P ? std::max(w->Pf.nblk, w->Pf.nwg()) : 0, kMaxVecBlocks}) *
                kMaxEpiReductions;
  w->part.alloc_zero(w->part_len, s);
  w->part2.alloc_zero(2 * kMaxVecBlocks, s);
  w->part_v.alloc_zero(kMaxVecBlocks, s);
  {
    // Persistent one-launch CG (cg_persist.hpp): bit-identical to the launch-per-kernel path, but NOT faster on
    // this GPU (a grid barrier costs what a kernel boundary costs: the L2 invalidate + the dependent-load chain
    // of the next phase; measured r01: 0.22 ms/iter either way on a config-5 problem with 16 workgroups, 2x slower
    // with one) => off unless asked for.  SCS_HIP_PERSIST = "W" or "WxG": W workgroups of G (1, 2, 4) 256-lane groups.
    const bool eligible = !w->At.has_slab && !w->Ar.has_slab && (!w->has_P || !w->Pf.has_slab) && !w->At.cs.ok && !w->Ar.cs.ok &&
                          (!w->has_P || !w->Pf.cs.ok) &&
                          2 * vec_blocks(l) + 2 * vec_blocks(std::max(n, m)) <= 2 * kMaxVecBlocks;
#ifdef SCS_HIP_LABS
    int wgs = 0, ng = 2;
    if (opts().persist_w > 0) {
      wgs = eligible ? std::max(0, std::min(opts().persist_w, kCgPersistMaxWgs)) : 0;
      const int b = opts().persist_g;
      if (b == 1 || b == 2 || b == 4) ng = b;
    }
    w->persist_wgs = wgs;
    w->persist_ng = ng;
    if (wgs > 0) w->persist_bar.alloc_zero(2, s);
#else
    (void)eligible;
#endif
  }
  w->sc.alloc_zero(S_COUNT, s);
  w->out.alloc_zero(256, s);
  w->fl.alloc_zero(F_COUNT, s);
  {
    std::vector<double> hh(l, 0.0);
    std::copy(w->c_orig.begin(), w->c_orig.end(), hh.begin());
    std::copy(w->b_orig.begin(), w->b_orig.end(), hh.begin() + n);
    w->h.upload(hh.data(), l, s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
  if (w->normalized) {
    w->scal.sigma = device_normalize_b_c(w->h, n, m, w->D, w->E, w->part, w->h_pin, s);
    std::vector<double> di(m), ei(n);
    for (int i = 0; i < m; ++i) di[i] = 1.0 / (w->scal.D[i] * w->scal.sigma);
    for (int i = 0; i < n; ++i) ei[i] = 1.0 / (w->scal.E[i] * w->scal.sigma);
    w->Dinv.upload(di.data(), m, s);
    w->Einv.upload(ei.data(), n, s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
  upload_cone_meta(w.get());
  {
    const double one = 1.0;
    HIP_CHECK(hipMemcpyAsync(w->sc.p + S_BOX_T, &one, sizeof(double), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
  }
  // ---- AA workspace ----
  w->aa.init(l, stgs->acceleration_lookback, stgs->acceleration_type_1, stgs->acceleration_regularization,
             stgs->acceleration_relaxation, /*safeguard_factor=*/1.0, /*max_weight_norm=*/1e10, s);
  mark("vectors, b/c scaling, cones, AA workspace");
  // ---- R, preconditioner (or G^{-1}), pre-solved g ----
  if (w->dense()) {
    w->dense_alloc();
#ifdef SCS_HIP_LABS
    w->persist_wgs = 0;
#endif
  }
  w->decide_k1dot(s);
  {
    // Round 5, late: the indirect path defers it too (SCS_HIP_LAZY_SETUP=0: inside scs_init) — its cold PCG for g is ~50 steps = 150 dependent
    // launches, three quarters of the dispatch chain of a small problem's scs_init; a batch runs it as ONE grouped cold solve (batch.hpp
    // apply_scale_updates, the path of an adaptive-scale update: bit-identical to the solo one), a lone workspace at its first solve.
    // (small problems only, n + m <= 32768: there the chain is what scs_init costs; a large problem keeps its cold solve out of scs_solve)
    const bool small_indirect = !w->dense() && (long)n + m <= 32768;
    w->setup_pending = (w->dense() || small_indirect) && opts().lazy_setup;
  }
  if (!w->setup_pending) {
    w->set_diag_r();
    w->update_work_cache();
  }
  HIP_CHECK(hipStreamSynchronize(s));
  mark("R, preconditioner, g = KKT^-1 h");
  w->setup_time = now_ms() - t0;
  return w.release();
}

