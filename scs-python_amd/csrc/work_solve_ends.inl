// work_solve_ends.inl — members of ScsHipWork (work.hpp): the two ends of a solve (begin_solve / finish_solve), shared by scs_solve and the grouped solve (batch.hpp)
  // ---- the two ends of a solve, shared by scs_solve and the grouped solve (batch.hpp) ----
  // per-solve state, info header and the initial iterate (cold: v = [0; 0; 1]; warm: from sol)
  void begin_solve(ScsSolution *sol, ScsInfo *info, int warm_start) {
    if (krylov_mode() == 1) mr_decide();  // (forced: from the first iteration, and named in the banner)
    std::memset(info, 0, sizeof(*info));
    info->setup_time = setup_time;
    if (dense())
      std::snprintf(info->lin_sys_solver, sizeof(info->lin_sys_solver), "dense-direct HIP gfx950 (explicit inverse of the reduced KKT matrix, order %d; fp64 MFMA Gauss-Jordan)", n);
    else if (persist_wgs > 0)
      std::snprintf(info->lin_sys_solver, sizeof(info->lin_sys_solver), "sparse-indirect HIP gfx950 (PCG, persistent %dx%d-wave kernel)",
                    persist_wgs, 4 * persist_ng);
    else {
      // what became of rows too long for the pass layout's count fields (A / A' / P): cut into pieces that ride in the passes, or
      // peeled off and summed from the plain CSR by the side launch
      const bool pieces = (Ar.cs.ok && Ar.cs.npieces > 0) || (At.cs.ok && At.cs.npieces > 0) || (has_P && Pf.cs.ok && Pf.cs.npieces > 0);
      const bool peeled = !pieces && ((Ar.cs.ok && Ar.npeel > 0) || (At.cs.ok && At.npeel > 0) || (has_P && Pf.cs.ok && Pf.npeel > 0));
      std::snprintf(info->lin_sys_solver, sizeof(info->lin_sys_solver), "sparse-indirect HIP gfx950 (%s SpMV%s, %s)",
                    At.cs.ok ? "column-sorted pass" : At.has_slab ? "L2-blocked slab" : "CSR-stream",
                    pieces ? ", long rows in pieces" : peeled ? ", long rows peeled" : "",
                    mr_active ? "MINRES, zero-cone block un-eliminated" : "PCG");
    }
    // per-solve state
    sum_log_scale_factor = 0; n_log_scale_factor = 0; last_scale_update_iter = 0; scale_updates = 0;
    rejected_accel = 0; accepted_accel = 0; aa_norm = 0;
    aa.reset(); aa.success = 0; aa.pending_safeguard = false; aa.st = ScsAaStats{};
    r = Residuals{};
    cg_res_min = 0;
    psd_res_min = 0;
    tot_cg_iters = 0;
    prof_ms[0] = prof_ms[1] = prof_ms[2] = 0;
    prof_n[0] = prof_n[1] = prof_n[2] = 0;
    prof_cone_ms = 0; prof_cone_n = 0;

    // ---- initial iterate ----
    {
      const double one = 1.0;
      if (warm_start) {
        // v = [x_hat; y_hat + s_hat / r_y; 1] with the normalised warm start (boundary work, O(l) on the host)
        std::vector<double> v0(l, 0.0);
        const double sg = normalized ? scal.sigma : 1.0;
        for (int i = 0; i < n; ++i) v0[i] = normalized ? sol->x[i] / (scal.E[i] / sg) : sol->x[i];
        for (int i = 0; i < m; ++i) {
          const double ry = (i < cone.z) ? 1.0 / (1000. * scale) : 1.0 / scale;
          const double yh = normalized ? sol->y[i] / (scal.D[i] / sg) : sol->y[i];
          const double sh = normalized ? sol->s[i] * (scal.D[i] * sg) : sol->s[i];
          v0[n + i] = yh + sh / ry;
        }
        for (long i = 0; i < l; ++i)
          if (!std::isfinite(v0[i])) v0[i] = 0.;
        v0[l - 1] = 1.0;
        HIP_CHECK(hipMemcpyAsync(v.p, v0.data(), sizeof(double) * l, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));  // v0 is a local
      } else {  // cold start: v = [0; 0; 1], nothing crosses PCIe
        HIP_CHECK(hipMemsetAsync(v.p, 0, sizeof(double) * l, stream));
        HIP_CHECK(hipMemcpyAsync(v.p + (l - 1), &one, sizeof(double), hipMemcpyHostToDevice, stream));
      }
      v_norm_fresh = false;
      HIP_CHECK(hipMemsetAsync(u.p, 0, sizeof(double) * l, stream));
      HIP_CHECK(hipMemcpyAsync(u.p + (l - 1), &one, sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    info->status_val = SCS_UNFINISHED;
  }
  // status, un-normalised (x, y, s) on the device and on the host, info; i = iterations done.  info->status_val holds
  // the verdict of the last convergence check (SCS_UNFINISHED: none fired).
  void finish_solve(ScsSolution *sol, ScsInfo *info, int i, double t_start, double t_lin, double t_cone, double t_acc,
                    bool grouped = false) {
    if (mr_active && !std::strstr(info->lin_sys_solver, "MINRES")) {  // the auto mode switched inside this solve
      char *at = std::strstr(info->lin_sys_solver, "PCG)");
      if (at) std::snprintf(at, sizeof(info->lin_sys_solver) - (size_t)(at - info->lin_sys_solver), "PCG, then MINRES)");
    }
    // ---- finalize ----
    const int max_iters = stgs.max_iters;
    if (!grouped) {  // (the grouped solve has read this problem's flags and residuals already)
      read_flags();
      populate_residuals(i == max_iters ? max_iters - 1 : i);  // loop ran out: rsk of the last iteration was computed
    }
    const double sg = normalized ? scal.sigma : 1.0;
    hipLaunchKernelGGL(k_unnormalize, dim3(vb((long)n + m)), dim3(kVecThreads), 0, stream, u.p, rsk.p,
                       normalized ? D.p : (const double *)nullptr, normalized ? E.p : (const double *)nullptr, sg, 1.0,
                       1.0, 1.0, n, m, solx.p, soly.p, sols.p);
    // complementary slackness s'y of the un-rescaled pair (fixed-order two-stage sum), then status and its scaling
    const int nbm = vb(m);
    hipLaunchKernelGGL(k_dot_part, dim3(nbm), dim3(kVecThreads), 0, stream, (const double *)sols.p, (const double *)soly.p, (long)m, part.p);
    std::vector<double> cs_part((size_t)nbm);
    HIP_CHECK(hipMemcpyAsync(cs_part.data(), part.p, sizeof(double) * nbm, hipMemcpyDeviceToHost, stream));
    info->iter = i;
    info->res_infeas = r.res_infeas;
    info->res_unbdd_a = r.res_unbdd_a;
    info->res_unbdd_p = r.res_unbdd_p;
    info->scale = scale;
    info->scale_updates = scale_updates;
    info->rejected_accel_steps = rejected_accel;
    info->accepted_accel_steps = accepted_accel;
    if (info->status_val == SCS_UNFINISHED) {
      if (r.tau > r.kap) info->status_val = SCS_SOLVED_INACCURATE;
      else if (r.bty_tau < r.ctx_tau) info->status_val = SCS_INFEASIBLE_INACCURATE;
      else info->status_val = SCS_UNBOUNDED_INACCURATE;
    }
    // final scaling on the device (a NaN factor marks a vector the status leaves undefined); the host copies are plain
    // downloads of the finished vectors, and the device copies stay behind for scs_hip_solution_to_device (scs/batch.py:
    // the RCCL gather starts from where the solutions live)
    double fx = 1., fy = 1., fs = 1.;
    switch (info->status_val) {
      case SCS_SOLVED:
      case SCS_SOLVED_INACCURATE:
        fx = fy = fs = safediv_pos(1.0, r.tau);
        info->gap = r.gap; info->res_pri = r.res_pri; info->res_dual = r.res_dual;
        info->pobj = r.xt_p_x / 2. + r.ctx;
        info->dobj = -r.xt_p_x / 2. - r.bty;
        std::snprintf(info->status, sizeof(info->status), "%s",
                      info->status_val == SCS_SOLVED ? "solved" : "solved (inaccurate - reached max_iters)");
        break;
      case SCS_INFEASIBLE:
      case SCS_INFEASIBLE_INACCURATE:
        fy = -1. / r.bty_tau;
        fx = fs = NAN;
        info->gap = info->res_pri = info->res_dual = NAN;
        info->pobj = INFINITY; info->dobj = INFINITY;
        std::snprintf(info->status, sizeof(info->status), "%s",
                      info->status_val == SCS_INFEASIBLE ? "infeasible" : "infeasible (inaccurate - reached max_iters)");
        break;
      case SCS_SIGINT:  // stopped by Ctrl-C: nothing is returned, as after a failure
        fx = fy = fs = NAN;
        info->gap = info->res_pri = info->res_dual = NAN;
        info->pobj = info->dobj = NAN;
        std::snprintf(info->status, sizeof(info->status), "interrupted");
        break;
      default:
        fx = fs = -1. / r.ctx_tau;
        fy = NAN;
        info->gap = info->res_pri = info->res_dual = NAN;
        info->pobj = -INFINITY; info->dobj = -INFINITY;
        std::snprintf(info->status, sizeof(info->status), "%s",
                      info->status_val == SCS_UNBOUNDED ? "unbounded" : "unbounded (inaccurate - reached max_iters)");
        break;
    }
    hipLaunchKernelGGL(k_scale3, dim3(vb((long)n + m)), dim3(kVecThreads), 0, stream, solx.p, soly.p, sols.p, n, m, fx, fy, fs);
    // (nothing is left running when scs_solve returns: a device-wide synchronize issued by the caller right after an
    // un-synchronised kernel was measured to take 25 ms on this runtime)
    download_solution(sol);
    sol_on_device = true;
    {
      double cs = 0.;
      for (double v : cs_part) cs += v;
      info->comp_slack = std::fabs(cs);
    }
    info->lin_sys_time = t_lin;
    info->cone_time = t_cone;
    info->accel_time = t_acc;
    info->cg_iters = (scs_int)tot_cg_iters;
    info->aa_stats = aa.st;
    info->solve_time = now_ms() - t_start;
  }
