// work_residuals.inl — members of ScsHipWork (work.hpp): residuals and termination tests, adaptive scale, Anderson acceleration glue, cone distances, the CSV log row
  // --------------------------------------------------------------- residuals
  void populate_residuals(int iter) {
    if (r.last_iter == iter) return;
    r.last_iter = iter;
    const double *x = u.p, *y = u.p + n, *s = rsk.p + n, *tau_ptr = u.p + (l - 1);
    // primal: 3 sums + 6 max over the A workgroups; dual: 4 sums + 6 max over the A' workgroups
    launch_spmv(Ar.view(), x, EpiResPri{s, h.p + n, normalized ? Dinv.p : nullptr, tau_ptr, y, part.p}, nullptr, stream);
    hipLaunchKernelGGL(k_fin_multi, dim3(1), dim3(kVecThreads), 0, stream, part.p, Ar.nwg(), 3, 6, out.p);
    if (has_P) launch_spmv(Pf.view(), x, EpiStore{px.p, 0}, nullptr, stream);
    launch_spmv(At.view(), y, EpiResDual{has_P ? px.p : nullptr, h.p, normalized ? Einv.p : nullptr, x, tau_ptr, part.p},
                nullptr, stream);
    hipLaunchKernelGGL(k_fin_multi, dim3(1), dim3(kVecThreads), 0, stream, part.p, At.nwg(), 4, 6, out.p + 16);
    HIP_CHECK(hipMemcpyAsync(h_pin, out.p, sizeof(double) * 32, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(h_pin + 32, u.p + (l - 1), sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(h_pin + 33, rsk.p + (l - 1), sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    consume_residuals(h_pin);
  }
  // host half of populate_residuals: res = the 32 reduced scalars of the two residual products, then u_tau, rsk_tau
  // (the grouped solve, batch.hpp, reads the records of all its problems with one copy and hands each one over here)
  void consume_residuals(const double *res) {
    const double pd = normalized ? scal.sigma * scal.sigma : 1.0;
    const double *hp = res, *hd = res + 16;
    r.tau = std::fabs(res[32]);
    r.kap_n = std::fabs(res[33]);
    r.kap = r.kap_n / pd;
    r.bty_tau_n = hp[RES_P_BTY];
    r.bty_tau = r.bty_tau_n / pd;
    r.sq_pri_n = hp[RES_P_SQ_N];
    r.sq_pri_o = hp[RES_P_SQ_O];
    r.nm_pri_n = hp[RES_P_MAX_N];
    r.nm_ax_s_btau = hp[RES_P_MAX_O];
    r.nm_ax_s = hp[RES_P_AXS_O];
    r.nm_ax = hp[RES_P_AX_O];
    r.nm_s = hp[RES_P_S_O];
    r.nm_ax_s_n = hp[RES_P_AXS_N];
    r.ctx_tau_n = hd[RES_D_CTX];
    r.ctx_tau = r.ctx_tau_n / pd;
    r.xt_p_x_tau_n = hd[RES_D_XPX];
    r.xt_p_x_tau = r.xt_p_x_tau_n / pd;
    r.sq_dual_n = hd[RES_D_SQ_N];
    r.sq_dual_o = hd[RES_D_SQ_O];
    r.nm_dual_n = hd[RES_D_MAX_N];
    r.nm_px_aty_ctau = hd[RES_D_MAX_O];
    r.nm_px = hd[RES_D_PX_O];
    r.nm_aty = hd[RES_D_ATY_O];
    r.nm_px_n = hd[RES_D_PX_N];
    r.nm_aty_n = hd[RES_D_ATY_N];
    r.bty = safediv_pos(r.bty_tau, r.tau);
    r.ctx = safediv_pos(r.ctx_tau, r.tau);
    r.xt_p_x = safediv_pos(r.xt_p_x_tau, r.tau * r.tau);
    r.gap = std::fabs(r.xt_p_x + r.ctx + r.bty);
    r.pobj = r.xt_p_x / 2. + r.ctx;
    r.dobj = -r.xt_p_x / 2. - r.bty;
    r.res_pri = safediv_pos(r.nm_ax_s_btau, r.tau);
    r.res_dual = safediv_pos(r.nm_px_aty_ctau, r.tau);
    r.res_unbdd_a = r.res_unbdd_p = r.res_infeas = NAN;
    if (r.ctx_tau < 0) {
      r.res_unbdd_a = safediv_pos(r.nm_ax_s, -r.ctx_tau);
      r.res_unbdd_p = safediv_pos(r.nm_px, -r.ctx_tau);
    }
    if (r.bty_tau < 0) r.res_infeas = safediv_pos(r.nm_aty, -r.bty_tau);
  }

  int has_converged(int iter) const {
    const double eps_abs = stgs.eps_abs, eps_rel = stgs.eps_rel, eps_infeas = stgs.eps_infeas;
    if (r.tau > 0.) {
      const double grl = std::max(std::max(std::fabs(r.xt_p_x), std::fabs(r.ctx)), std::fabs(r.bty));
      const double prl = std::max(std::max(nm_b_orig * r.tau, r.nm_s), r.nm_ax) / r.tau;
      const double drl = std::max(std::max(nm_c_orig * r.tau, r.nm_px), r.nm_aty) / r.tau;
      if (std::isless(r.res_pri, eps_abs + eps_rel * prl) && std::isless(r.res_dual, eps_abs + eps_rel * drl) &&
          std::isless(r.gap, eps_abs + eps_rel * grl))
        return SCS_SOLVED;
    }
    if (std::isless(r.res_unbdd_a, eps_infeas) && std::isless(r.res_unbdd_p, eps_infeas) && iter > 0) return SCS_UNBOUNDED;
    if (std::isless(r.res_infeas, eps_infeas) && iter > 0) return SCS_INFEASIBLE;
    return 0;
  }

  // the adaptive-scale rule on the residuals in `r` (host state only).  true: `scale` changed — the caller rebuilds
  // R, the preconditioner and g, resets the acceleration and re-expresses v (apply_scale_update; batch.hpp does the
  // same for a sub-list of its group)
  bool decide_scale_update(int iter) {
    const int since = iter - last_scale_update_iter;
    const double rel_pri = safediv_pos(r.nm_ax_s_btau, std::max(std::max(r.nm_ax, r.nm_s), nm_b_orig * r.tau));
    const double rel_dual = safediv_pos(r.nm_px_aty_ctau, std::max(std::max(r.nm_px, r.nm_aty), nm_c_orig * r.tau));
    sum_log_scale_factor += std::log(rel_pri) - std::log(rel_dual);
    n_log_scale_factor++;
    const double factor = std::sqrt(std::exp(sum_log_scale_factor / (double)n_log_scale_factor));
    if (since < 100) return false;
    const double new_scale = std::min(std::max(scale * factor, 1e-4), 1e6);
    if (new_scale == scale) return false;
    if (factor > std::sqrt(10.) || factor < 1. / std::sqrt(10.)) {
      scale_updates++;
      sum_log_scale_factor = 0;
      n_log_scale_factor = 0;
      last_scale_update_iter = iter;
      scale = new_scale;
      return true;
    }
    return false;
  }
  // A spinning kernel timed out (another process holds part of the GPU): put the workspace back where scs_solve found it, as far as
  // that is possible — the scale and what hangs on it, the device scalars and flags, cold cone workspaces (the eigenvectors of
  // earlier solves are gone: a first solve restarts bit for bit, a later one from a cold projection) — and never spin again.
  void spin_fallback(double scale_entry) {
    (void)hipStreamSynchronize(stream);
    (void)hipGetLastError();
    psd_mc_cap = 0;
#ifdef SCS_HIP_LABS
    if (persist_wgs > 0) { persist_wgs = 0; graphs_ready = false; }
#endif
    stall = nullptr;
    stall_fl = nullptr;
    in_capture = false;
    HIP_CHECK(hipMemsetAsync(fl.p, 0, sizeof(int) * F_COUNT, stream));
    std::memset(h_flags, 0, sizeof(int) * F_COUNT);
    for (auto &hf : h_flags_slot) if (hf) std::memset(hf, 0, sizeof(int) * F_COUNT);
    HIP_CHECK(hipMemsetAsync(sc.p, 0, sizeof(double) * S_COUNT, stream));
    const double one = 1.0;
    HIP_CHECK(hipMemcpyAsync(sc.p + S_BOX_T, &one, sizeof(double), hipMemcpyHostToDevice, stream));
    if (psd_scratch.p) HIP_CHECK(hipMemsetAsync(psd_scratch.p, 0, sizeof(double) * psd_scratch.n, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    scale = scale_entry;
    set_diag_r();
    update_work_cache();
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  void update_scale(int iter) {
    if (!decide_scale_update(iter)) return;
    set_diag_r();
    update_work_cache();
    aa.reset();  // reset acceleration
    hipLaunchKernelGGL(k_v_rescale, dim3(vb(l)), dim3(kVecThreads), 0, stream, v.p, rsk.p, u.p, ut.p, diag_r.p, l);
    v_norm_fresh = false;
  }

  // --------------------------------------------------------------------- AA
  void aa_apply() {  // f = v (map output), x = v_prev (map input)
    aa_norm = 0;
    if (aa.mem <= 0) return;
    // acceleration_interval == 1: the verdict of the previous step's safeguard has not been read yet (it rides with the
    // CG flags of the NEXT linear solve) — a rejected step must reset the history before it is extended
    if (aa.pending_safeguard) read_flags();
    aa_norm = aa.apply(v.p, v_prev.p);
    if (aa.success) v_norm_fresh = false;
  }

  void aa_safeguard() {  // f_new = v, x_new = v_prev
    if (aa.mem <= 0) return;
    if (!aa.safeguard(v.p, v_prev.p, fl.p + F_SAFE_BAD)) { accepted_accel++; return; }
    v_norm_fresh = false;
  }

  // ||v - Pi(v)||_2 for a host vector in ORIGINAL units: Pi = projection onto K (dual = 0) or K* (dual = 1), with the
  // hot-path cone kernels (footer diagnostics of a verbose solve: "dist(s, K)", "dist(y, K*)").  The box-cone warm start
  // is saved and restored; PSD eigenvector warm starts are not used (and are left as the projection leaves them).
  double cone_dist(const double *hv, int dual) {
    if (!std::isfinite(hv[0])) return NAN;
    // scratch: rsk (recomputed by every iteration that needs it) holds the vector, tmp_m its projection
    HIP_CHECK(hipMemcpyAsync(rsk.p, hv, sizeof(double) * m, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(tmp_m.p, rsk.p, sizeof(double) * m, hipMemcpyDeviceToDevice, stream));
    double box_t = 1.0;
    HIP_CHECK(hipMemcpyAsync(&box_t, sc.p + S_BOX_T, sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    {
      // the caller's (unscaled) box bounds and a cold PSD start for this one projection; put back whatever happens
      // (a refused launch throws out of project_nonlinear_cones)
      struct Restore {
        ScsHipWork *w;
        int warm;
        explicit Restore(ScsHipWork *w_) : w(w_), warm(w_->psd_warm) {
          std::swap(w->box_bl.p, w->box_bl_orig.p);
          std::swap(w->box_bu.p, w->box_bu_orig.p);
          w->psd_warm = 0;
        }
        ~Restore() {
          w->psd_warm = warm;
          std::swap(w->box_bl.p, w->box_bl_orig.p);
          std::swap(w->box_bu.p, w->box_bu_orig.p);
        }
      } restore(this);
      if (cone.z + cone.l > 0)
        hipLaunchKernelGGL(k_proj_zl, dim3(ceil_div(cone.z + cone.l, kConeThreads)), dim3(kConeThreads), 0, stream, tmp_m.p, cone.z, cone.l, dual);
      project_nonlinear_cones(tmp_m.p, dual);
    }
    const int nb = vb(m);
    hipLaunchKernelGGL(k_aa_diffsq, dim3(nb), dim3(kVecThreads), 0, stream, (const double *)rsk.p, (const double *)tmp_m.p, (long)m, part.p);
    std::vector<double> hp(nb);
    HIP_CHECK(hipMemcpyAsync(hp.data(), part.p, sizeof(double) * nb, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(sc.p + S_BOX_T, &box_t, sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    double ss = 0.;
    for (double v : hp) ss += v;
    return std::sqrt(ss);
  }

  // one CSV row: residuals of this iteration are already in `r`; diff norms are reduced here
  void log_csv_row(FILE *f, int iter, double elapsed_ms) {
    const int nbl = vb(l);
    hipLaunchKernelGGL(k_diff_norms, dim3(nbl), dim3(kVecThreads), 0, stream, u.p, ut.p, v.p, v_prev.p, l, part.p);
    hipLaunchKernelGGL(k_fin_multi, dim3(1), dim3(kVecThreads), 0, stream, part.p, nbl, 2, 2, out.p + 40);
    HIP_CHECK(hipMemcpyAsync(h_pin + 40, out.p + 40, sizeof(double) * 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    write_csv_row(f, iter, r, scale, h_pin + 40, aa_norm, elapsed_ms / 1e3);
  }

