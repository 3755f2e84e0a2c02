// work_admm.inl — members of ScsHipWork (work.hpp): the steps of an ADMM iteration as enqueue functions, the run-ahead queue and its stall
// recovery, (labs) graph capture, project_lin_sys and the cone projections
  // ------------------------------------------------------------ ADMM steps
  void set_iter_params(int iter, int slot = 0) {
    h_params = h_params_base + slot * P_COUNT;
    d_params = d_params_base + slot * P_COUNT;
    h_params[P_DO_SCALE] = iter >= 1 ? 1.0 : 0.0;
    h_params[P_RES_MIN] = cg_res_min;  // residuals of the last convergence CHECK (not of a logging-only evaluation)
    h_params[P_IPOW] = std::pow((double)iter + 1, 1.5);
    h_params[P_FIRST] = iter < 1 ? 1.0 : 0.0;
    h_params[P_PSD_TOL2] = psd_tol2_for(iter);
    const bool dbg_tol = (opts().debug & DBG_TOL) != 0;  // SCS_HIP_DEBUG=tol (tools/dbg/run_ahead_tol.py)
    if (dbg_tol) std::fprintf(stderr, "[scs-hip] iter %d slot %d: res_min %.17g psd level %.3e tol2 %.3e\n", iter, slot, cg_res_min, psd_res_min, h_params[P_PSD_TOL2]);
  }
  // everything of project_lin_sys up to (and including) the fused, warm-started CG start
  void enqueue_lin_sys_head() {
    const int nbl = vb(l);
    // (the sum-of-squares partials of v are in part_v: enqueue_v_update of the previous iteration or ensure_v_norm)
    hipLaunchKernelGGL(k_prep, dim3(nbl), dim3(kVecThreads), 0, stream, v.p, v_prev.p, ut.p, ws.p, u.p, g.p, diag_r.p, n, m,
                       d_params, part_v.p, nbl, sc.p, part2.p, stall);
    // y0 = v_y + R_y^{-1} A ws   (start of the y recurrence, lives in ut_y)
    launch_spmv(Ar.view(), ws.p, EpiY{ut.p + n, rdy(), v.p + n}, stall, stream);
    // r0 = R_x (v_x - ws) - P ws - A' y0 ; p0 = M r0 ; partials for ||r0||_inf and r0'M r0
    if (has_P) launch_spmv(Pf.view(), ws.p, EpiStore{cg_Gp.p, 0}, stall, stream);
    launch_spmv(At.view(), ut.p + n, EpiR0{cg_r.p, cg_p.p, cg_M.p, rdx(), v.p, ws.p, has_P ? cg_Gp.p : nullptr, part.p},
                stall, stream);
    // tolerance, ||r0||, r0'M r0, step counter, zero-rhs short circuit: one finalize launch
    hipLaunchKernelGGL(k_fin_head, dim3(1), dim3(kVecThreads), 0, stream, part2.p, nbl, part.p, At.nwg(), d_params, sc.p, fl.p,
                       ut.p, (long)n + m, stall);
    // (k1dot: the first step's alpha needs sum r_x p0^2 of the p0 = M r0 the start has just formed)
#ifdef SCS_HIP_LABS
    if (k1dot) hipLaunchKernelGGL(k_pp_part, dim3(vb(n)), dim3(kVecThreads), 0, stream, (const double *)cg_p.p, rdx(), n, part_pp.p, stall);
#endif
    if (mr_active) enqueue_mr_start();
  }
  // dense direct variant of the linear solve of an iteration: rhs = R_x v_x - A' v_y;  u~_x = G^{-1} rhs;  u~_y = v_y + R_y^{-1} A u~_x.
  // Three dependent launches behind k_prep, no convergence flag: nothing here (or behind it) waits for the device.
  void enqueue_lin_sys_dense() {
    const int nbl = vb(l);
    hipLaunchKernelGGL(k_prep, dim3(nbl), dim3(kVecThreads), 0, stream, v.p, v_prev.p, ut.p, ws.p, u.p, g.p, diag_r.p, n, m,
                       d_params, part_v.p, nbl, sc.p, part2.p, stall);
    launch_spmv(At.view(), v.p + n, EpiDenseRhs{cg_b.p, rdx(), v.p}, stall, stream);
    dense_gemv(cg_b.p, ut.p, stall);
    launch_spmv(Ar.view(), ut.p, EpiY{ut.p + n, rdy(), v.p + n}, stall, stream);
  }
  // ||v||^2 partials for k_prep when something other than enqueue_v_update wrote v (start, AA, scale update)
  void ensure_v_norm() {
    if (v_norm_fresh) return;
    hipLaunchKernelGGL(k_sumsq, dim3(vb(l)), dim3(kVecThreads), 0, stream, v.p, l, part_v.p);
    v_norm_fresh = true;
  }
#ifdef SCS_HIP_LABS
  // small-problem variant: same normalisation / warm start, then ONE launch for tolerance, CG start and CG loop
  void enqueue_lin_sys_persist() {
    std::unique_ptr<SpinLink> link;
    if (!in_capture) link.reset(new SpinLink(this));  // (a captured launch is replayed outside any chain: SCS_HIP_PERSIST is a lab switch)
    const int nbl = vb(l);
    hipLaunchKernelGGL(k_prep, dim3(nbl), dim3(kVecThreads), 0, stream, v.p, v_prev.p, ut.p, ws.p, u.p, g.p, diag_r.p, n, m,
                       d_params, part_v.p, nbl, sc.p, part2.p, stall);
    CgPersistArgs a{};
    a.Ar = Ar.view().csr; a.At = At.view().csr;
    if (has_P) a.Pf = Pf.view().csr;
    a.has_P = has_P ? 1 : 0; a.n = n; a.m = m;
    a.diag_r = diag_r.p; a.v = v.p; a.ws = ws.p; a.ut = ut.p;
    a.r = cg_r.p; a.p = cg_p.p; a.Gp = cg_Gp.p; a.z = tmp_m.p; a.M = cg_M.p;
    a.part = part.p; a.part2 = part2.p + 2 * nbl; a.part_p = part2.p; a.np_p = nbl;
    a.params = d_params; a.sc = sc.p; a.fl = fl.p; a.max_its = 10 * n; a.bar = persist_bar.p;
    if (persist_ng == 4)
      hipLaunchKernelGGL(k_cg_persist<4>, dim3(persist_wgs), dim3(4 * kVecThreads), cg_persist_lds<4>(), stream, a);
    else if (persist_ng == 2)
      hipLaunchKernelGGL(k_cg_persist<2>, dim3(persist_wgs), dim3(2 * kVecThreads), cg_persist_lds<2>(), stream, a);
    else
      hipLaunchKernelGGL(k_cg_persist<1>, dim3(persist_wgs), dim3(kVecThreads), cg_persist_lds<1>(), stream, a);
    enqueue_flag_readback();
  }
  void finish_lin_sys_persist() {
    sync_flags();
    if (h_flags[F_PERSIST_ERR]) throw SpinTimeout("persistent CG kernel: grid barrier timed out");
    last_cg_iters = h_flags[F_ITERS];
    note_cg_iters(last_cg_iters);
    tot_cg_iters += last_cg_iters;
  }
#endif
  // tau (the y block is already in ut_y: it was carried along the CG recurrence)
  void enqueue_lin_sys_tail() {
    const int nb1 = vb(l - 1);
    hipLaunchKernelGGL(k_tau_dots, dim3(nb1), dim3(kVecThreads), 0, stream, ut.p, v.p, g.p, diag_r.p, l - 1, part.p, stall_fl);
  }
  void enqueue_cones() {  // (tau is formed in k_cone_pre's prologue from the k_tau_dots partials)
    hipLaunchKernelGGL(k_cone_pre, dim3(vb(l)), dim3(kVecThreads), 0, stream, ut.p, u.p, v.p, g.p, n, m, cone.z, cone.l,
                       d_params, sc.p, part.p, vb(l - 1), diag_r.p, stall_fl);
    psd_tol2 = d_params + P_PSD_TOL2;
    project_nonlinear_cones(u.p + n, 1);
    psd_tol2 = nullptr;
  }
  void enqueue_v_update() {
    hipLaunchKernelGGL(k_v_update, dim3(vb(l)), dim3(kVecThreads), 0, stream, v.p, u.p, ut.p, stgs.alpha, l, part_v.p, stall);
    v_norm_fresh = true;
  }

  // dense direct linsys: a plain iteration has nothing the host must look at (no CG flags): enqueue it and go on — the queue only
  // drains at Anderson steps and convergence checks
  void enqueue_plain_dense(int iter) {
    set_iter_params(iter, iter & 1);
    ensure_v_norm();
    enqueue_lin_sys_dense();
    enqueue_lin_sys_tail();
    enqueue_cones();
    enqueue_v_update();
    last_cg_iters = 0;
  }

  // ---- run-ahead mode: one whole plain iteration (no convergence check, no AA, no logging) in the queue ----
  // head + CG chunk + tau/cones/v update + flag copy + event; nothing here waits for the device.
  // queue_empty: nothing of an earlier iteration is still in the queue.  Only then may the Krylov method change: a
  // stalled iteration i is finished by run_cg(mode 2) with the method of the workspace, and a switch made while i + 1 was being
  // enqueued would continue i's PCG recurrence with MINRES steps that never had their start.
  void enqueue_plain_iteration(int iter, bool queue_empty) {
    const int slot = iter & 1;
    set_iter_params(iter, slot);
    ensure_v_norm();
    if (queue_empty) mr_decide();
    stall = fl.p + F_STALL;
    stall_fl = fl.p;
    enqueue_lin_sys_head();
    // the largest step count of the last 8 solves + 1 (the newest count is one iteration stale here; is_plain caps it):
    // an unused step costs four ~1-2 us launches, a stall a drained queue and a host round trip (~100 us)
    int chunk = std::max(2, recent_cg_max() + 1);
    if (pipe_chunk_override > 0) chunk = pipe_chunk_override;
    prof_step[slot] = -1;
    for (int k = 0; k < chunk; ++k) {
      if (mr_active) {
        enqueue_mr_step(k);
      } else if (profile && k == chunk / 2) {  // one CG step of the queued iteration bracketed by events: nothing waits for them here
        for (auto &e : ev_prof[slot]) if (!e) HIP_CHECK(hipEventCreate(&e));
        enqueue_cg_step(ut.p, ut.p + n, ev_prof[slot]);
        prof_step[slot] = k;
      } else {
        enqueue_cg_step(ut.p, ut.p + n);
      }
    }
    if (mr_active) enqueue_mr_finish();
    enqueue_lin_sys_tail();
    cone_sampled[slot] = false;
    if (profile) {  // the nonlinear cone projections of this queued iteration between two events (read when it is finished)
      for (auto &e : ev_cone[slot]) if (!e) HIP_CHECK(hipEventCreate(&e));
      hipLaunchKernelGGL(k_cone_pre, dim3(vb(l)), dim3(kVecThreads), 0, stream, ut.p, u.p, v.p, g.p, n, m, cone.z, cone.l,
                         d_params, sc.p, part.p, vb(l - 1), diag_r.p, stall_fl);
      HIP_CHECK(hipEventRecord(ev_cone[slot][0], stream));
      psd_tol2 = d_params + P_PSD_TOL2;
      project_nonlinear_cones(u.p + n, 1);
      psd_tol2 = nullptr;
      HIP_CHECK(hipEventRecord(ev_cone[slot][1], stream));
      cone_sampled[slot] = true;
    } else {
      enqueue_cones();
    }
    enqueue_v_update();
    stall = nullptr;
    stall_fl = nullptr;
    HIP_CHECK(hipMemcpyAsync(h_flags_slot[slot], fl.p, sizeof(int) * F_COUNT, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipEventRecord(ev_iter[slot], stream));
  }
  // Wait for iteration `iter` of the run-ahead queue.  Returns false if its CG chunk was too short: the rest of that
  // iteration and everything queued behind it did nothing; the caller finishes the iteration synchronously.
  // SCS_HIP_DEBUG=pipe: per-iteration CG step counts and run-ahead stalls on stderr.
  static bool debug_pipe() {
    return (opts().debug & DBG_PIPE) != 0;
  }
  // host wait for an event: SCS_HIP_WAIT=block -> hipEventSynchronize, spin -> poll hipEventQuery (lab knob)
  static int wait_mode() {
    return opts().wait_spin ? 1 : 0;  // (labs knob)
  }
  static void wait_event(hipEvent_t e) {
    if (wait_mode() == 1) {
      for (;;) {
        const hipError_t q = hipEventQuery(e);
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) HIP_CHECK(q);
        __builtin_ia32_pause();
      }
    }
    HIP_CHECK(hipEventSynchronize(e));
  }
  bool finish_plain_iteration(int iter) {
    const int slot = iter & 1;
    HIP_CHECK(hipGetLastError());  // a refused launch (hipLaunchKernelGGL reports nothing) surfaces here, once per iteration
    wait_event(ev_iter[slot]);
    const int *hf = h_flags_slot[slot];
    if (hf[F_STALL]) {
      ++pipe_stalls;
      if (debug_pipe()) std::fprintf(stderr, "[scs-hip] iter %d: STALL after %d CG steps\n", iter, hf[F_ITERS]);
      return false;
    }
    std::memcpy(h_flags, hf, sizeof(int) * F_COUNT);
    process_pending_flags();  // e.g. the verdict of the Anderson safeguard enqueued in the iteration before
    last_cg_iters = hf[F_ITERS];
    note_cg_iters(last_cg_iters);
    tot_cg_iters += last_cg_iters;
    if (debug_pipe()) std::fprintf(stderr, "[scs-hip] iter %d: %d CG steps (queued ahead)\n", iter, last_cg_iters);
    if (profile && cone_sampled[slot]) {
      float c = 0;
      if (hipEventElapsedTime(&c, ev_cone[slot][0], ev_cone[slot][1]) == hipSuccess) { prof_cone_ms += c; prof_cone_n++; }
    }
    if (profile && prof_step[slot] >= 0 && last_cg_iters > prof_step[slot]) note_cg_sample(ev_prof[slot]);  // the sampled step really ran
    return true;
  }
  // after a stall: drain the queue, lower the flags and finish iteration `iter` the synchronous way
  void recover_stalled_iteration(int iter) {
    HIP_CHECK(hipStreamSynchronize(stream));
    std::memcpy(h_flags, h_flags_slot[iter & 1], sizeof(int) * F_COUNT);
    h_flags[F_DONE] = 0;
    h_flags[F_STALL] = 0;
    const int zeros[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(fl.p + F_DONE, &zeros[0], sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(fl.p + F_STALL, &zeros[1], sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    set_iter_params(iter, iter & 1);
    run_cg(ut.p, ws.p, 10 * n, 2);  // continues from the intact CG state (the flags say how far it got)
    enqueue_lin_sys_tail();
    enqueue_cones();
    enqueue_v_update();
    v_norm_fresh = true;
  }

#ifdef SCS_HIP_LABS
  hipGraphExec_t capture(const std::function<void()> &body) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    in_capture = true;
    try {
      body();
    } catch (...) {
      in_capture = false;
      (void)hipStreamEndCapture(stream, &graph);
      if (graph) (void)hipGraphDestroy(graph);
      throw;
    }
    in_capture = false;
    HIP_CHECK(hipStreamEndCapture(stream, &graph));
    HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphDestroy(graph));
    return exec;
  }
  void build_graphs() {
    if (graphs_ready || !graphs_enabled) return;
    const bool keep_fresh = v_norm_fresh;  // capturing enqueues nothing: host-side state must not move
    if (persist_wgs > 0) g_pre[0] = capture([&] { enqueue_lin_sys_persist(); });
    for (int i = 0; i < kNumGraphs && persist_wgs == 0; ++i) {
      const int c = kGraphSteps[i];
      g_pre[i] = capture([&] {
        enqueue_lin_sys_head();
        for (int k = 0; k < c; ++k) enqueue_cg_step(ut.p, ut.p + n);
        enqueue_flag_readback();
      });
      g_cg[i] = capture([&] {
        for (int k = 0; k < c; ++k) enqueue_cg_step(ut.p, ut.p + n);
        enqueue_flag_readback();
      });
    }
    g_post = capture([&] {
      enqueue_lin_sys_tail();
      enqueue_cones();
      enqueue_v_update();
    });
    v_norm_fresh = keep_fresh;
    graphs_ready = true;
  }
#else
  void build_graphs() {}
#endif

  void project_lin_sys(int iter, bool graph) {
    // The parameter block is host memory the kernels read in place, and the cones of the previous iteration (enqueued, not
    // waited for) read P_PSD_TOL2 from this slot.  It moves after convergence checks (the stream is idle then) and ONCE
    // more, when the residual-tied level is switched off (psd_tol2_for): wait before overwriting it.  (Run-ahead
    // iterations alternate between two slots instead: enqueue_plain_iteration.)
    if (n_psd + n_cs > 0 && psd_tol2_for(iter) != h_params_base[P_PSD_TOL2]) HIP_CHECK(hipStreamSynchronize(stream));
    mr_decide();
    set_iter_params(iter);
    ensure_v_norm();
    if (dense()) {
      enqueue_lin_sys_dense();
      last_cg_iters = 0;
      return;
    }
#ifdef SCS_HIP_LABS
    if (persist_wgs > 0) {
      if (graph) HIP_CHECK(hipGraphLaunch(g_pre[0], stream));
      else enqueue_lin_sys_persist();
      finish_lin_sys_persist();
      return;
    }
#endif
    if (kLabsBuild && graph && !mr_active) {
      int gi = 0;
      const int want = std::max(1, std::min(last_cg_iters + 2, kGraphSteps[kNumGraphs - 1]));
      while (gi + 1 < kNumGraphs && kGraphSteps[gi] < want) ++gi;  // smallest captured chunk that covers `want`
      HIP_CHECK(hipGraphLaunch(g_pre[gi], stream));
      sync_flags();
      run_cg(ut.p, ws.p, 10 * n, 2);
    } else {
      enqueue_lin_sys_head();
      run_cg(ut.p, ws.p, 10 * n, 1);  // the CG start is already enqueued: continue eagerly
    }
  }

  // in-place projection of the m-slice y onto K (dual=0) or K* (dual=1), rows z/l excluded (handled by caller)
  void project_nonlinear_cones(double *y, int dual) {
    if (cone.bsize > kBoxMultiMin) {  // large box cone: one launch per Newton round over many workgroups (cones.hpp)
      if (!box_parts.p) { box_parts.alloc_zero(2 * kBoxMultiMaxWgs, stream); box_ticket.alloc_zero(1, stream); }
      const int wgs = box_multi_wgs(cone.bsize);
      double *state = sc.p + S_BOX_T;  // {t (warm start of the next call), stop flag}
      for (int round = 0; round < kBoxRounds; ++round)
        hipLaunchKernelGGL(k_proj_box_round, dim3(wgs), dim3(kBoxMultiThreads), 0, stream, (const double *)(y + cone.off_box), box_bl.p,
                           box_bu.p, cone.bsize, state, box_parts.p, box_ticket.p, dual, round, stall);
      hipLaunchKernelGGL(k_proj_box_apply, dim3(wgs), dim3(kBoxMultiThreads), 0, stream, y + cone.off_box, box_bl.p, box_bu.p, cone.bsize,
                         state, dual, stall);
    } else if (cone.bsize > 0) {
      hipLaunchKernelGGL(k_proj_box, dim3(1), dim3(kBoxThreads), 0, stream, y + cone.off_box, box_bl.p, box_bu.p, cone.bsize,
                         sc.p + S_BOX_T, dual, stall);
    }
    // short SOCs + small PSD matrices (nothing big of either kind): one launch for both (psd.hpp k_proj_soc_psd_small)
    const bool soc_psd_fused = soc_psd_one_launch && n_soc > 0 && n_soc_big == 0 && n_psd > 0 && n_psd_big == 0 && !psd_small_one_wave;
    if (soc_psd_fused) {
      const int sb = soc_wave_blocks(n_soc, soc_G);
      hipLaunchKernelGGL(k_proj_soc_psd_small, dim3(sb + n_psd), dim3(kPsdSmallThreads), 0, stream, y, soc_off.p, soc_dim.p, n_soc, soc_G, sb,
                         PsdBatch{psd_off.p, psd_order.p, psd_woff.p, n_psd}, psd_scratch.p, psd_warm, stall, psd_tol2);
    } else if (n_soc > 0) {  // self-dual
      hipLaunchKernelGGL(k_proj_soc_wave, dim3(soc_wave_blocks(n_soc, soc_G)), dim3(kConeThreads), 0, stream, y,
                         soc_off.p, soc_dim.p, n_soc, soc_G, stall);
      if (n_soc_big > 0)
        hipLaunchKernelGGL(k_proj_soc_block, dim3(n_soc_big), dim3(kConeThreads), 0, stream, y, soc_off.p, soc_dim.p,
                           soc_big.p, n_soc_big, stall);
    }
    if (n_psd > 0 && !soc_psd_fused) launch_psd(y, psd_off.p, psd_order.p, psd_woff.p, n_psd, n_psd_big);  // self-dual
    if (n_cs > 0) {  // Hermitian PSD: self-dual
      CsBatch C{cs_off.p, cs_order.p, cs_soff.p, n_cs};
      hipLaunchKernelGGL(k_cs_expand, dim3(n_cs), dim3(256), 0, stream, y, C, cs_stage.p, stall);
      launch_psd(cs_stage.p, cs_poff.p, cs_porder.p, cs_woff.p, n_cs, n_cs_big);
      hipLaunchKernelGGL(k_cs_extract, dim3(n_cs), dim3(256), 0, stream, y, C, cs_stage.p, stall);
    }
    if (cone.ep > 0)  // K = K_exp: dual -> project onto K_exp^*
      hipLaunchKernelGGL(k_proj_exp, dim3(ceil_div(cone.ep, kConeThreads)), dim3(kConeThreads), 0, stream, y + cone.off_ep,
                         cone.ep, dual ? 0 : 1, stall);
    if (cone.ed > 0)  // K = K_exp^*: dual -> project onto K_exp
      hipLaunchKernelGGL(k_proj_exp, dim3(ceil_div(cone.ed, kConeThreads)), dim3(kConeThreads), 0, stream, y + cone.off_ed,
                         cone.ed, dual ? 1 : 0, stall);
    if (!cone.p.empty()) {
      const int np = (int)cone.p.size();
      if (dual)
        hipLaunchKernelGGL(k_proj_pow_dual, dim3(ceil_div(np, kConeThreads)), dim3(kConeThreads), 0, stream, y + cone.off_p,
                           pow_a.p, np, stall);
      else
        hipLaunchKernelGGL(k_proj_pow_primal, dim3(ceil_div(np, kConeThreads)), dim3(kConeThreads), 0, stream, y + cone.off_p,
                           pow_a.p, np, stall);
    }
  }

