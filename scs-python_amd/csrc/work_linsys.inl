// work_linsys.inl — members of ScsHipWork (work.hpp): the linear solve of an ADMM iteration — dense direct path, the K1 / K2 / K3 products, PCG
// (fused start, steps, run_cg), the labs-only Krylov variants, the standalone KKT solve and g = KKT^-1 h
  // -------------------------------------------------------------- helpers
  int vb(long nelem) const { return vec_blocks(nelem); }

  // ---- dense direct linsys (dense.hpp; linsys == 1): G^{-1} = (R_x + P + A' R_y^{-1} A)^{-1} resident in HBM, rebuilt whenever R changes
  int linsys = 0;  // 0: indirect (PCG), 1: dense direct
  DevBuf<double> dn_G, dn_Pk, dn_L, dn_Rt, dn_part;
  int dn_NP = 0, dense_factorisations = 0;
  bool dense() const { return linsys == 1; }
  DenseMat dense_mat() const { return DenseMat{dn_G.p, dn_Pk.p, dn_L.p, dn_Rt.p, n, dn_NP}; }
  DenseSrc dense_src() const {
    return DenseSrc{At.rowptr.p, At.col.p, At.val.p, Ar.rowptr.p, Ar.col.p, Ar.val.p, has_P ? Pf.rowptr.p : nullptr,
                    has_P ? Pf.col.p : nullptr, has_P ? Pf.val.p : nullptr, diag_r.p};
  }
  void dense_alloc() {
    dn_NP = dense_np(n);
    dn_G.alloc((size_t)dn_NP * dn_NP);
    dn_Pk.alloc((size_t)kDenseB * kDenseB);
    dn_L.alloc((size_t)dn_NP * kDenseB);
    dn_Rt.alloc((size_t)dn_NP * kDenseB);
    dn_part.alloc_zero(dense_symv_part_len(dn_NP), stream);
  }
  // x = X' b over the WHOLE computed inverse X (default), or SCS_HIP_DENSE_GEMV=half: the two-launch product that reads only the tiles on
  // and below the diagonal.  The half product is NOT the default although it halves the bytes of the HBM-bound part of a batch: a
  // Gauss-Jordan inverse is accurate on ONE side (here || X G - I || ~ eps kappa, so X' b solves G x = b to ~ kappa eps), while its
  // asymmetry — what a product that mirrors one triangle sees — is kappa times larger: measured on the KKT test systems
  // (kappa = 2e4) 1e-11 against 3.5e-8 relative error (tools/dbg/dense_gemv_err.py, profiles/r04_dense_linsys.txt).
  static bool dense_full_gemv() {
    return opts().dense_full_gemv;  // (labs: SCS_HIP_DENSE_GEMV=half)
  }
  void dense_gemv(const double *b, double *x, const int *st) {
    if (dense_full_gemv())
      hipLaunchKernelGGL(k_dense_gemv, dim3(dense_gemv_blocks(n)), dim3(kDenseThreads), 0, stream, (const double *)dn_G.p, dn_NP, n, b, x, st);
    else
      dense_apply(dn_G.p, dn_NP, n, b, dn_part.p, x, st, stream);
  }
  // Dense workspaces finish their setup — R, G^{-1}, g = KKT^{-1} [c; -b] — at the first solve (or update) instead of inside scs_init:
  // a batch of them then forms and inverts all its matrices in ONE batched sweep (GroupSolve::run), 66 launches for the whole group
  // instead of 66 launch-bound ones per member (SCS_HIP_LAZY_SETUP=0: inside scs_init).
  bool setup_pending = false, setup_failed = false;
  std::string setup_failed_msg(int member = -1) const {
    return std::string("hip_dense: the inverse of the reduced KKT matrix is not finite") +
           (member >= 0 ? " (member " + std::to_string(member) + " of the batch)" : "") +
           " (a vanishing pivot block — column-rank-deficient A with a tiny rho_x?); use LinearSolver.HIP_INDIRECT for this problem";
  }
  void finish_pending_setup() {
    if (setup_failed) throw std::runtime_error(setup_failed_msg());  //  a failed setup stays failed: no solve on a non-finite inverse
    if (!setup_pending) return;
    const double t0 = now_ms();
    set_diag_r();
    update_work_cache();
    HIP_CHECK(hipStreamSynchronize(stream));
    setup_pending = false;
    setup_time += now_ms() - t0;
    if (dense()) {
      //  the Gauss-Jordan sweep of the dense path does not pivot and checks nothing on the way: at least the solve it has just
      // been used for, g = KKT^-1 [c; -b], must be finite (g' R g is on the device already: one double)
      double gg = 0.;
      HIP_CHECK(hipMemcpyAsync(&gg, sc.p + S_GG, sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      if (!std::isfinite(gg)) {
        setup_failed = true;
        throw std::runtime_error(setup_failed_msg());
      }
    }
  }
  void dense_refactor() {  // diag_r must be current on the stream
    dense_factor(dense_src(), dense_mat(), stream);
    ++dense_factorisations;
  }

  void set_diag_r() {
    diag_r_structured = true;
    hipLaunchKernelGGL(k_set_diag_r, dim3(vb(l)), dim3(kVecThreads), 0, stream, diag_r.p, n, m, cone.z, stgs.rho_x, scale);
    if (dense()) { dense_refactor(); return; }
    hipLaunchKernelGGL(k_precond, dim3(vb(n)), dim3(kVecThreads), 0, stream, At.rowptr.p, At.col.p, At.val.p, diag_r.p,
                       has_P ? Pdiag.p : (const double *)nullptr, cg_M.p, n);
    if (mr_ready) mr_precond();
  }

  // R_x / R_y as the SpMV epilogues take them: two scalars inside the ADMM workspace (set_diag_r built diag_r), the
  // vector for the standalone KKT entry point (arbitrary diag_r)
  // (not when the iteration is replayed from captured hipGraphs — SCS_HIP_PIPELINE=0: kernel arguments are frozen at
  // capture and `scale` changes with every adaptive scale update; the vector is updated in place)
  bool diag_r_structured = false;
  bool r_scalars() const { return diag_r_structured && pipelined; }
  RDiag rdx() const { return r_scalars() ? RDiag(stgs.rho_x, stgs.rho_x, 0) : RDiag(diag_r.p); }
  RDiag rdy() const { return r_scalars() ? RDiag(1.0 / (1000. * scale), 1.0 / scale, cone.z) : RDiag(diag_r.p + n); }
  // p'Gp from K1 instead of K2 (cg_k1dot.hpp), for large LPs / SOCPs whose A and A' both use the column-sorted pass layout.  OPT-IN
  // (SCS_HIP_K1DOT=1): measured on the metric workload it makes K2 3-4 us faster (90.0 -> 86.4 us: K2 = K1) but the iteration 1.7 % SLOWER
  // (310-312 -> 305-306 iters/s, steady window 507-511 -> 491-496): the second reduction chain (r_x p^2 through k_cg_dir -> k_cg_update's
  // prologue, one more pass behind the CG start) and K1's block reduction cost more than K2's 16 MB of p saved.
#ifdef SCS_HIP_LABS
  bool k1dot = false;
  DevBuf<double> part_k1, part_pp;
  void decide_k1dot(hipStream_t s) {
    k1dot = opts().k1dot && !has_P && At.cs.ok && Ar.cs.ok && persist_wgs == 0;
    if (k1dot) {
      part_k1.alloc_zero((size_t)std::max(Ar.nwg(), 1) * kMaxEpiReductions, s);
      part_pp.alloc_zero((size_t)kMaxVecBlocks, s);
    }
  }
#else
  static constexpr bool k1dot = false;
  void decide_k1dot(hipStream_t) {}
#endif
  // Gp = (R_x + P + A' R_y^{-1} A) x ; partial p.Gp into part[0..At.nblk)
  // step_counter != nullptr marks the A product of a CG step (its workgroup 0 advances the step parity)
  // second half of Gp when A' has the split layout (EpiGp::split): Gp = cg_Gp + gp2()
  double *gp2() const { return At.cs.ok && At.cs.split > 1 && !At.cs.combine() ? At.cs_part1.p : nullptr; }
#ifdef SCS_HIP_LABS
  // the two products of a CG step on the k1dot path: z = R_y^{-1} A p with the partials of (A p)'z, then the raw A'z (cg_Gp [+ gp2()])
  void matvec_k1dot(const double *x, const int *done, int *step_counter, hipEvent_t *evs = nullptr) {
    if (evs) HIP_CHECK(hipEventRecord(evs[0], stream));
    launch_spmv(Ar.view(), x, EpiDivRDot{tmp_m.p, rdy(), part_k1.p}, done, stream, step_counter);
    if (evs) HIP_CHECK(hipEventRecord(evs[1], stream));
    launch_spmv(At.view(), tmp_m.p, EpiAtRaw{cg_Gp.p, gp2()}, done, stream);
    if (evs) HIP_CHECK(hipEventRecord(evs[2], stream));
  }
#endif
  void matvec(const double *x, const int *done, int *step_counter = nullptr) {
    launch_spmv(Ar.view(), x, EpiDivR{tmp_m.p, rdy()}, done, stream, step_counter);
    if (has_P) launch_spmv(Pf.view(), x, EpiStore{cg_Gp.p, 0}, done, stream);
    launch_spmv(At.view(), tmp_m.p, EpiGp{cg_Gp.p, x, rdx(), has_P ? 1 : 0, part.p, gp2()}, done, stream);
  }

  void read_flags() {
    HIP_CHECK(hipGetLastError());  // launches are not checked one by one: a refused one is caught here
    HIP_CHECK(hipMemcpyAsync(h_flags, fl.p, sizeof(int) * F_COUNT, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    process_pending_flags();
  }

  void process_pending_flags() {
    if (h_flags[F_PERSIST_ERR]) throw SpinTimeout("a spinning multi-workgroup kernel (persistent CG / PSD sweeps) timed out at its barrier");
    if (aa.pending_safeguard) {
      const bool bad = h_flags[F_SAFE_BAD] != 0;
      aa.safeguard_verdict(bad);
      if (bad) rejected_accel++;
      else accepted_accel++;
    }
  }

  // ---- enqueue helpers (used both eagerly and under stream capture) ----
  void enqueue_cg_start(double *xout, const double *warm) {
    const int nb = vb(n);
    if (warm) matvec(warm, nullptr);
    hipLaunchKernelGGL(k_cg_init, dim3(nb), dim3(kVecThreads), 0, stream, cg_b.p, cg_Gp.p, warm, cg_M.p, xout, cg_r.p, cg_p.p,
                       n, warm ? 1 : 0, fl.p, part.p, (const double *)gp2());
    hipLaunchKernelGGL(k_fin_cg_init, dim3(1), dim3(kVecThreads), 0, stream, part.p, nb, 0, sc.p, fl.p);
    HIP_CHECK(hipMemsetAsync(fl.p + F_ITERS, 0, sizeof(int), stream));
#ifdef SCS_HIP_LABS
    if (k1dot) hipLaunchKernelGGL(k_pp_part, dim3(vb(n)), dim3(kVecThreads), 0, stream, (const double *)cg_p.p, rdx(), n, part_pp.p, (const int *)nullptr);
#endif
  }
  // yacc != nullptr: carry y += alpha R_y^{-1} A p along (ADMM path, see k_prep).  evs: three events around the two products (in-situ
  // kernel timing of one step: bench.py's roofline)
  void enqueue_cg_step(double *xout, double *yacc, hipEvent_t *evs = nullptr) {
    const int nb = vb(std::max(n, yacc ? m : 0));
#ifdef SCS_HIP_LABS
    if (k1dot) {
      matvec_k1dot(cg_p.p, fl.p + F_DONE, fl.p + F_STEP, evs);
      hipLaunchKernelGGL(k_cg_update_k1dot, dim3(nb), dim3(kVecThreads), 0, stream, xout, cg_r.p, (const double *)cg_p.p, (const double *)cg_Gp.p,
                         (const double *)gp2(), (const double *)cg_M.p, n, yacc, (const double *)tmp_m.p, m, (const double *)part_k1.p, Ar.nwg(),
                         (const double *)part_pp.p, vb(n), rdx(), sc.p, (const int *)fl.p, part2.p);
      hipLaunchKernelGGL(k_cg_dir_pp, dim3(vb(n)), dim3(kVecThreads), 0, stream, cg_p.p, (const double *)cg_r.p, (const double *)cg_M.p, n,
                         (const double *)part2.p, nb, rdx(), part_pp.p, sc.p, fl.p);
      return;
    }
#endif
    if (evs) {
      HIP_CHECK(hipEventRecord(evs[0], stream));
      launch_spmv(Ar.view(), cg_p.p, EpiDivR{tmp_m.p, rdy()}, fl.p + F_DONE, stream, fl.p + F_STEP);
      HIP_CHECK(hipEventRecord(evs[1], stream));
      if (has_P) {
        launch_spmv(Pf.view(), cg_p.p, EpiStore{cg_Gp.p, 0}, fl.p + F_DONE, stream);
        HIP_CHECK(hipEventRecord(evs[3], stream));
      }
      launch_spmv(At.view(), tmp_m.p, EpiGp{cg_Gp.p, cg_p.p, rdx(), has_P ? 1 : 0, part.p, gp2()}, fl.p + F_DONE, stream);
      HIP_CHECK(hipEventRecord(evs[2], stream));
    } else
    matvec(cg_p.p, fl.p + F_DONE, fl.p + F_STEP);
    if (cg_fuse()) {  // small systems: update + direction as one launch (vec.hpp k_cg_update_dir; same bits)
      hipLaunchKernelGGL(k_cg_update_dir, dim3(nb), dim3(kVecThreads), 0, stream, xout, cg_r.p, cg_p.p, (const double *)cg_Gp.p,
                         (const double *)cg_M.p, n, yacc, (const double *)tmp_m.p, m, (const double *)part.p, At.nwg(), sc.p, fl.p, part2.p,
                         (const double *)gp2(), cg_ticket.p);
      return;
    }
    hipLaunchKernelGGL(k_cg_update, dim3(nb), dim3(kVecThreads), 0, stream, xout, cg_r.p, cg_p.p, cg_Gp.p, cg_M.p, n, yacc,
                       tmp_m.p, m, part.p, At.nwg(), sc.p, fl.p, part2.p, (const double *)gp2());
    hipLaunchKernelGGL(k_cg_dir, dim3(vb(n)), dim3(kVecThreads), 0, stream, cg_p.p, cg_r.p, cg_M.p, n, part2.p, nb, sc.p, fl.p);
  }
  // SCS_HIP_CG_FUSE=0: always two launches
  bool cg_fuse_on = opts().cg_fuse;  // (labs switch) read when the workspace is made
  bool cg_fuse() const { return cg_fuse_on && n <= kCgFuseMaxN; }
  void enqueue_flag_readback() {
    HIP_CHECK(hipMemcpyAsync(h_flags, fl.p, sizeof(int) * F_COUNT, hipMemcpyDeviceToHost, stream));
  }
  void sync_flags() {
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(stream));
    process_pending_flags();
  }

  // ---- MINRES on the system with the zero-cone block un-eliminated (minres.hpp), the second Krylov method of the indirect solve.
  // SCS_HIP_KRYLOV = cg (default) | minres (whenever the cone has zero rows) | auto (switches a workspace over, for good, once the third
  // quartile of its last 8 PCG solves exceeds kMrAutoSteps steps and z >= kMrAutoZ).  Cold KKT solves (init, scale updates) stay with PCG.
  // NOT the default, by measurement (round 5, profiles/r05_config3_minres.txt): on BASELINE config 3 — the case it was built for, 10 % zero-cone
  // rows, PCG at 170 steps per ADMM iteration over a whole solve — MINRES needs 241 steps per iteration at the same stopping rule and
  // 825 instead of 700 ADMM iterations: 29.1 s against 13.9 s.  The round-4 prototype compared the two from a RANDOM warm start (1.7 x
  // fewer steps); inside the ADMM loop the warm start is the previous iterate, the residual has to fall by a modest factor only, and the
  // reduced residual — which MINRES does not minimise — first rises.  The recursion's residual equals the true one (SCS_HIP_MR_CHECK).
#ifdef SCS_HIP_LABS
  static constexpr int kMrAutoSteps = 96, kMrAutoZ = 256;
  int krylov = opts().krylov;  // 0 cg, 1 minres (whenever z > 0), 2 auto; read when the workspace is made
  int krylov_mode() const { return krylov; }
  bool mr_active = false, mr_ready = false, mr_allowed = true;
  double mr_tolf = opts().mr_tolf;  // (lab) MINRES stops at mr_tolf x the PCG tolerance
  long mr_N = 0;
  int mr_nred = 1;
  DevBuf<double> mr_B, mr_YP, mr_W, mr_d, mr_rho, mr_Minv, mr_Y, mr_sc, mr_partA, mr_partB, mr_partV, mr_partR, mr_zval;
  DevBuf<int> mr_zptr, mr_zidx;
  void mr_precond() {
    hipLaunchKernelGGL(k_mr_precond_x, dim3(ceil_div(n, kVecThreads)), dim3(kVecThreads), 0, stream, At.rowptr.p, At.col.p, At.val.p, rdx(), rdy(),
                       has_P ? Pdiag.p : (const double *)nullptr, n, cone.z, mr_Minv.p);
    hipLaunchKernelGGL(k_mr_precond_z, dim3(ceil_div(cone.z, kVecThreads)), dim3(kVecThreads), 0, stream, Ar.rowptr.p, Ar.col.p, Ar.val.p, rdy(), n,
                       cone.z, mr_Minv.p);
  }
  void mr_setup() {  // once per workspace, at the switch (a host round trip for the prefix sums of A_z')
    if (mr_ready) return;
    const long N = (long)n + cone.z;
    mr_N = N;
    mr_B.alloc_zero((size_t)(3 * N), stream);
    mr_YP.alloc_zero((size_t)(2 * N), stream);
    mr_W.alloc_zero((size_t)(3 * N), stream);
    mr_d.alloc_zero((size_t)N, stream);
    mr_rho.alloc_zero((size_t)N, stream);
    mr_Minv.alloc_zero((size_t)N, stream);
    mr_Y.alloc_zero((size_t)N, stream);
    mr_sc.alloc_zero(kMrScalars, stream);
    mr_partA.alloc_zero(part_len, stream);
    mr_partB.alloc_zero(part_len, stream);
    mr_partV.alloc_zero(kMaxVecBlocks, stream);
    mr_partR.alloc_zero(kMaxVecBlocks, stream);
    mr_nred = std::max(1, std::min(kMaxVecBlocks, ceil_div(n, kVecThreads)));
    DevBuf<int> cnt;
    cnt.alloc((size_t)n);
    hipLaunchKernelGGL(k_mr_azt_count, dim3(ceil_div(n, kVecThreads)), dim3(kVecThreads), 0, stream, At.rowptr.p, At.col.p, n, cone.z, cnt.p);
    std::vector<int> hc((size_t)n), hp((size_t)n + 1, 0);
    cnt.download(hc.data(), (size_t)n, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    for (int j = 0; j < n; ++j) hp[(size_t)j + 1] = hp[(size_t)j] + hc[(size_t)j];
    mr_zptr.upload(hp.data(), hp.size(), stream);
    mr_zidx.alloc((size_t)std::max(hp[(size_t)n], 1));
    mr_zval.alloc((size_t)std::max(hp[(size_t)n], 1));
    hipLaunchKernelGGL(k_mr_azt_fill, dim3(ceil_div(n, kVecThreads)), dim3(kVecThreads), 0, stream, At.rowptr.p, At.col.p, At.val.p, n, mr_zptr.p,
                       mr_zidx.p, mr_zval.p);
    mr_precond();
    HIP_CHECK(hipStreamSynchronize(stream));  // hp, hc are locals
    mr_ready = true;
  }
  // decided where an ADMM iteration's linear solve is enqueued (never inside one)
  bool mr_precond_stale = false;  // a grouped solve changed the scale behind MINRES's back (batch.hpp): refreshed at the next decision
  void mr_decide() {
    if (mr_ready && mr_precond_stale && mr_allowed) { mr_precond(); mr_precond_stale = false; }
    if (mr_active || !mr_allowed || cone.z <= 0 || dense() || persist_wgs > 0 || k1dot || in_capture) return;
    const int mode = krylov_mode();
    if (mode == 0) return;
    if (mode == 2 && !(cone.z >= kMrAutoZ && recent_cg_q3() > kMrAutoSteps)) return;
    mr_setup();
    mr_active = true;
    graphs_ready = false;  // (captured CG chunks are of no use any more; graphs are not rebuilt for MINRES)
  }
  void enqueue_mr_start() {  // behind enqueue_lin_sys_head: cg_r holds r0, the flags and the tolerance are set
    const long N = mr_N;
    hipLaunchKernelGGL(k_mr_init, dim3(vb(N)), dim3(kVecThreads), 0, stream, (const double *)cg_r.p, (const double *)mr_Minv.p, n, N, mr_B.p, mr_B.p + N,
                       mr_YP.p, mr_W.p, mr_W.p + N, mr_d.p, mr_rho.p, mr_partV.p, stall);
    hipLaunchKernelGGL(k_mr_fin0, dim3(1), dim3(kVecThreads), 0, stream, (const double *)mr_partV.p, vb(N), mr_sc.p, stall);
  }
  void enqueue_mr_step(int k) {
    const long N = mr_N;
    const int bank = k & 1;
    double *r1 = mr_B.p + (k % 3) * N, *r2 = mr_B.p + ((k + 1) % 3) * N, *r3 = mr_B.p + ((k + 2) % 3) * N;
    double *yp = mr_YP.p + (k & 1) * N, *ypn = mr_YP.p + ((k + 1) & 1) * N;
    double *w1 = mr_W.p + (k % 3) * N, *w2 = mr_W.p + ((k + 1) % 3) * N, *wn = mr_W.p + ((k + 2) % 3) * N;
    const double *bk = mr_sc.p + kMrBank0 + kMrBankLen * bank;
    const int *done = fl.p + F_DONE;
    launch_spmv(Ar.view(), yp, EpiMrU{tmp_m.p, mr_Y.p, yp, r1, bk, rdy(), n, cone.z, mr_partA.p}, done, stream);
    if (has_P) launch_spmv(Pf.view(), yp, EpiStore{mr_Y.p, 0}, done, stream);
    launch_spmv(At.view(), tmp_m.p, EpiMrY{mr_Y.p, gp2(), yp, r1, bk, rdx(), has_P ? 1 : 0, mr_partB.p}, done, stream);
    hipLaunchKernelGGL(k_mr_v1, dim3(vb(N)), dim3(kVecThreads), 0, stream, (const double *)mr_Y.p, (const double *)gp2(), (const double *)r2,
                       (const double *)mr_Minv.p, n, N, r3, ypn, (const double *)mr_partA.p, Ar.nwg(), (const double *)mr_partB.p, At.nwg(), mr_sc.p, bank,
                       mr_partV.p, (const int *)fl.p);
    hipLaunchKernelGGL(k_mr_v2, dim3(vb(N)), dim3(kVecThreads), 0, stream, (const double *)yp, (const double *)w1, (const double *)w2, wn, mr_d.p, mr_rho.p,
                       (const double *)r3, N, (const double *)mr_partV.p, vb(N), mr_sc.p, bank, (const int *)fl.p);
    hipLaunchKernelGGL(k_mr_red, dim3(mr_nred), dim3(kVecThreads), 0, stream, (const int *)mr_zptr.p, (const int *)mr_zidx.p, (const double *)mr_zval.p,
                       (const double *)mr_rho.p, n, rdy(), mr_partR.p, (const int *)fl.p);
    hipLaunchKernelGGL(k_mr_fin, dim3(1), dim3(kVecThreads), 0, stream, (const double *)mr_partR.p, mr_nred, sc.p, fl.p, mr_tolf);
  }
  void enqueue_mr_finish() {  // x = ws + d_x, y = v_y + R_y^{-1} A x.  Idempotent: after a run-ahead stall it simply runs again
    hipLaunchKernelGGL(k_mr_x, dim3(vb(n)), dim3(kVecThreads), 0, stream, ut.p, (const double *)ws.p, (const double *)mr_d.p, n, (const int *)fl.p, stall);
    launch_spmv(Ar.view(), ut.p, EpiY{ut.p + n, rdy(), v.p + n}, stall, stream);
  }
#else
  // (the product's Krylov method is PCG; MINRES lives in the labs build.  The names the loop uses fold to nothing here.)
  static constexpr bool mr_active = false, mr_ready = false;
  static constexpr int krylov_mode() { return 0; }
  void mr_precond() {}
  void mr_decide() {}
  void enqueue_mr_start() {}
  void enqueue_mr_step(int) {}
  void enqueue_mr_finish() {}
#endif

  // PCG on cg_b (rhs, length n); solution accumulates in xout.  S_TOL / F_DONE must be set on device.
  // Returns CG iterations taken.  `started` = the CG start (and `done_iters` steps) were already enqueued
  // and synced by a captured graph.
  // mode 0: enqueue the CG start here; 1: the start is already enqueued (not synced);
  // 2: start + some steps were enqueued by a captured graph and the flags are synced.
  int run_cg(double *xout, const double *warm, int max_its, int mode = 0) {
    int done_iters = 0;
    const bool started = mode == 2;
    if (mode == 0) {
      enqueue_cg_start(xout, warm);
    } else if (mode == 2) {
      done_iters = h_flags[F_ITERS];
      if (h_flags[F_DONE] || done_iters >= max_its) {
        last_cg_iters = done_iters;
        if (xout == ut.p) note_cg_iters(done_iters);
        tot_cg_iters += done_iters;
        if (mr_active && xout == ut.p) enqueue_mr_finish();
        return done_iters;
      }
    }
    const bool use_graph = started && xout == ut.p && graphs_ready && !mr_active;  // graphs are captured for the ADMM buffers only (PCG steps)
    double *yacc = (xout == ut.p) ? ut.p + n : nullptr;  // ADMM path carries the y block along the recurrence
    int chunk = started ? std::max(2, std::min(std::max(done_iters / 2, 4), 64)) : std::max(1, std::min(last_cg_iters + 2, 64));  // a host round trip costs ~30 us, an unused CG step four ~1 us launches
    while (true) {
      const int iters_before = done_iters;
      if (use_graph) {
        int gi = 0;
        while (gi + 1 < kNumGraphs && kGraphSteps[gi + 1] <= chunk) ++gi;
        HIP_CHECK(hipGraphLaunch(g_cg[gi], stream));
        sync_flags();
      } else {
        const int sample_it = chunk / 2;  // a mid-chunk step: not the one right behind the host sync
        for (int it = 0; it < chunk; ++it) {
          if (mr_active && xout == ut.p) enqueue_mr_step(done_iters + it);
          else enqueue_cg_step(xout, yacc, (profile && it == sample_it) ? ev : nullptr);
        }
        read_flags();
        if (profile && h_flags[F_ITERS] - iters_before > sample_it) note_cg_sample(ev);  // the sampled step really ran
      }
      done_iters = h_flags[F_ITERS];
      if (h_flags[F_DONE] || done_iters >= max_its) break;
      chunk = std::max(2, std::min(std::max(done_iters / 2, 4), 64));
    }
    last_cg_iters = done_iters;
    if (xout == ut.p) note_cg_iters(done_iters);  // (not the cold KKT solves of init / scale updates)
    tot_cg_iters += done_iters;
    if (mr_active && xout == ut.p) {
      enqueue_mr_finish();
      const bool check = opts().mr_check;  // lab: the TRUE reduced residual of the x MINRES returned
      if (check && !has_P) {
        launch_spmv(At.view(), ut.p + n, EpiR0{cg_r.p, cg_p.p, cg_M.p, rdx(), v.p, ut.p, nullptr, part.p}, nullptr, stream);
        std::vector<double> hr((size_t)n);
        double hs[S_COUNT];
        cg_r.download(hr.data(), (size_t)n, stream);
        sc.download(hs, S_COUNT, stream);
        HIP_CHECK(hipStreamSynchronize(stream));
        double mx = 0;
        for (double x : hr) mx = std::max(mx, std::fabs(x));
        std::fprintf(stderr, "[scs-hip] MINRES %d steps: recursion |r_red|_inf %.3e, true %.3e, tol %.3e\n", done_iters, hs[S_RNORM], mx, hs[S_TOL]);
      }
    }
    return done_iters;
  }

  // standalone KKT solve on a device vector rhs (length n+m), cold start, fixed tolerance
  int kkt_solve(double *rhs, double tol) {
    hipLaunchKernelGGL(k_kkt_prep, dim3(vb(m)), dim3(kVecThreads), 0, stream, rhs, diag_r.p, tmp_m.p, n, m);
    launch_spmv(At.view(), tmp_m.p, EpiRhs{cg_b.p, rhs}, nullptr, stream);
    HIP_CHECK(hipMemsetAsync(part.p, 0, sizeof(double), stream));
    hipLaunchKernelGGL(k_fin_tol, dim3(1), dim3(kVecThreads), 0, stream, part.p, 1, 0.0, 1.0, tol, 0, (const double *)nullptr, sc.p,
                       fl.p);
    int its = 0;
    if (dense()) dense_gemv(cg_b.p, ws.p, nullptr);
    else its = run_cg(ws.p, nullptr, 10 * n);  // solution in ws
    launch_spmv(Ar.view(), ws.p, EpiStore{tmp_m.p, 0}, nullptr, stream);
    hipLaunchKernelGGL(k_kkt_y, dim3(vb(m)), dim3(kVecThreads), 0, stream, rhs, tmp_m.p, diag_r.p, n, m);
    HIP_CHECK(hipMemcpyAsync(rhs, ws.p, sizeof(double) * n, hipMemcpyDeviceToDevice, stream));
    return its;
  }

  // g = (R + M)^{-1} [c; -b];  cache g'Rg
  void update_work_cache() {
    hipLaunchKernelGGL(k_g_rhs, dim3(vb((long)n + m)), dim3(kVecThreads), 0, stream, g.p, h.p, n, m);
    kkt_solve(g.p, 1e-12);
    const int nb = vb(l - 1);
    hipLaunchKernelGGL(k_gg, dim3(nb), dim3(kVecThreads), 0, stream, g.p, diag_r.p, l - 1, part.p);
    hipLaunchKernelGGL(k_fin_store_sum, dim3(1), dim3(kVecThreads), 0, stream, part.p, nb, sc.p, (int)S_GG);
  }

