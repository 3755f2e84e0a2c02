// loop.hpp — scs_solve: the ADMM loop (solve_impl), the Ctrl-C listener, verbose banner / table / footer
// (one of the units csrc/scs_hip.hip is assembled from — ONE translation unit, in this order: runtime.hpp, device_csr.hpp, work.hpp
// [+ work_linsys.inl, work_admm.inl, work_residuals.inl, work_solve_ends.inl], io.hpp, setup.hpp, loop.hpp, batch.hpp, the C ABI in scs_hip.hip,
// lab_entries.hpp; split out of the 3 800-line file of rounds 1-5 in round 6 — VERDICT r05 item 6 — without moving a line of code)
#pragma once
// ================================================================ solve
static void fill_nan(double *p, long nelem) {
  for (long i = 0; i < nelem; ++i) p[i] = NAN;
}

// ---- Ctrl-C: the reference builds its core with -DCTRLC=1 (R:meson.build:118) and reports SCS_SIGINT = -5, "interrupted"
// (R:scs/py/__init__.py:20).  While at least one solve runs, SIGINT is caught here (the previous disposition — Python's handler — comes
// back when the last one returns); every ADMM loop looks at the flag once per iteration and stops with that status and NaN vectors,
// as a failure does.  A ctypes / glue call into a multi-second device loop is otherwise uninterruptible.  SCS_HIP_CTRLC=0: hands off.
struct InterruptListener {
  static std::atomic<int> &flag() { static std::atomic<int> f{0}; return f; }
  static void on_sigint(int) { flag().store(1, std::memory_order_relaxed); }
  static bool enabled() { static const bool on = [] { const char *e = getenv("SCS_HIP_CTRLC"); return !(e && e[0] == '0'); }(); return on; }
  static std::mutex &mtx() { static std::mutex m; return m; }
  static int &users() { static int u = 0; return u; }
  static struct sigaction &saved() { static struct sigaction sa; return sa; }
  InterruptListener() {
    if (!enabled()) return;
    std::lock_guard<std::mutex> g(mtx());
    if (users()++ == 0) {
      flag().store(0);
      struct sigaction sa;
      std::memset(&sa, 0, sizeof(sa));
      sa.sa_handler = on_sigint;
      sigemptyset(&sa.sa_mask);
      sa.sa_flags = SA_RESTART;  //  other threads' blocking system calls are restarted, not failed with EINTR
      sigaction(SIGINT, &sa, &saved());
    }
  }
  ~InterruptListener() {
    if (!enabled()) return;
    std::lock_guard<std::mutex> g(mtx());
    if (--users() == 0) {
      // put the previous disposition back only if ours is still the installed one: a handler the application installed while the
      // solve was running is not overwritten
      struct sigaction cur;
      if (sigaction(SIGINT, nullptr, &cur) == 0 && cur.sa_handler == on_sigint) sigaction(SIGINT, &saved(), nullptr);
    }
  }
  static bool interrupted() { return enabled() && flag().load(std::memory_order_relaxed) != 0; }
};

static scs_int solve_impl(ScsHipWork *w, ScsSolution *sol, ScsInfo *info, scs_int warm_start) {
  std::lock_guard<std::mutex> lock(w->mtx);
  InterruptListener ctrlc;
  HIP_CHECK(hipSetDevice(w->device));
  const int n = w->n, m = w->m;
  const long l = w->l;
  hipStream_t s = w->stream;
  w->finish_pending_setup();  // the deferred end of scs_init: counted in setup_time, so the clock of the solve starts behind it
  const double t_start = now_ms();
  w->begin_solve(sol, info, warm_start);
  double t_lin = 0, t_cone = 0, t_acc = 0;
  FILE *csv = nullptr;
  if (!w->log_csv_filename.empty()) {
    csv = std::fopen(w->log_csv_filename.c_str(), "w");
    if (csv) std::fputs(kCsvHeader, csv);
  }
  const bool verbose = w->stgs.verbose != 0;
  if (verbose) {
    std::printf("------------------------------------------------------------------\n");
    std::printf("\t  scs-hip v%s - MI355X-native Splitting Conic Solver path\n", scs_version());
    std::printf("------------------------------------------------------------------\n");
    std::printf("problem:  variables n: %d, constraints m: %d\n", n, m);
    std::printf("cones: \t  z: %d, l: %d, box: %d, q: %zu, s: %zu, cs: %zu, ep: %d, ed: %d, p: %zu\n", w->cone.z, w->cone.l,
                w->cone.bsize, w->cone.q.size(), w->cone.s.size(), w->cone.cs.size(), w->cone.ep, w->cone.ed,
                w->cone.p.size());
    std::printf("settings: eps_abs: %.1e, eps_rel: %.1e, eps_infeas: %.1e\n\t  alpha: %.2f, scale: %.2e, adaptive_scale: %d\n"
                "\t  max_iters: %d, normalize: %d, rho_x: %.2e\n\t  acceleration_lookback: %d, acceleration_interval: %d\n",
                w->stgs.eps_abs, w->stgs.eps_rel, w->stgs.eps_infeas, w->stgs.alpha, w->scale, w->stgs.adaptive_scale,
                w->stgs.max_iters, w->stgs.normalize, w->stgs.rho_x, w->stgs.acceleration_lookback,
                w->stgs.acceleration_interval);
    std::printf("lin-sys:  %s\n\t  nnz(A): %ld, nnz(P): %ld\n", info->lin_sys_solver, w->At.nnz, w->has_P ? w->Pf.nnz : 0L);
    std::printf("------------------------------------------------------------------\n");
    std::printf(" iter | pri res | dua res |   gap   |   obj   |  scale  | time (s)\n");
    std::printf("------------------------------------------------------------------\n");
  }

  int i;
  const int max_iters = w->stgs.max_iters;
  // hipGraphs pay off when the iteration is launch/latency-bound (measured 8-14 % at l <= 1e4, nothing at
  // l >= 3e5) and cost ~0.1 s to capture: build them lazily, only for small problems and long solves.
  const long graph_max_l = opts().graph_max_l;  // (labs)
  const bool graphs_wanted = w->graphs_enabled && !w->profile && l <= graph_max_l && !w->pipelined && !w->dense();
  bool use_graphs = graphs_wanted && w->graphs_ready;
  const bool run_ahead = w->pipelined && w->persist_wgs == 0 && !w->dense();  // (in-situ profiling samples ride along: enqueue_plain_iteration)
  // an iteration is "plain" when the host has nothing to decide in it: no convergence check / print / log row,
  // no Anderson step, not the last one.  Plain iterations may be enqueued whole, and one ahead (run-ahead mode).
  auto is_plain = [&](int it) {
    if (it <= 0 || it >= max_iters - 1 || csv || it == w->mark_iter) return false;
    if (it % 25 == 0 || (verbose && it % 250 == 0)) return false;
    if (w->aa.mem > 0 && it % w->stgs.acceleration_interval == 0) return false;
    if (w->last_cg_iters > 120) return false;  // very long linear solves: enqueue them in adaptive chunks as before
    return true;
  };
  int enq_upto = -1;  // run-ahead: last iteration already in the queue
  w->mark_ms = -1;
  for (i = 0; i < max_iters; ++i) {
    if (InterruptListener::interrupted()) {
      info->status_val = SCS_SIGINT;
      break;
    }
    if (i == w->mark_iter) {
      HIP_CHECK(hipStreamSynchronize(s));
      w->mark_ms = now_ms() - t_start;
      w->mark_cg = w->tot_cg_iters;
      w->mark_aa_calls = w->aa.st.iter;
      w->mark_aa_accept = w->aa.st.n_accept;
    }
    if (w->dense() && w->pipelined && is_plain(i)) {  // nothing to wait for: the queue drains at the next Anderson step / check
      double t = now_ms();
      w->enqueue_plain_dense(i);
      if ((i & 63) == 0) HIP_CHECK(hipGetLastError());
      t_lin += now_ms() - t;
      continue;
    }
    if (run_ahead && (enq_upto >= i || is_plain(i))) {  // (already queued: is_plain may have changed its mind since)
      double t = now_ms();
      if (enq_upto < i) { w->enqueue_plain_iteration(i, true); enq_upto = i; }
      if (is_plain(i + 1) && enq_upto < i + 1) { w->enqueue_plain_iteration(i + 1, false); enq_upto = i + 1; }
      if (!w->finish_plain_iteration(i)) {
        w->recover_stalled_iteration(i);
        enq_upto = i;  // whatever was queued behind the stall did nothing
      }
      t_lin += now_ms() - t;
      continue;
    }
    if (graphs_wanted && !use_graphs && i == 64) {
      w->build_graphs();
      use_graphs = w->graphs_ready;
    }
    const bool aa_now = w->aa.mem > 0 && i > 0 && (i % w->stgs.acceleration_interval == 0);
    double t = now_ms();
    if (aa_now) {
      w->aa_apply();
      t_acc += now_ms() - t;
    }
    const bool check = (i % 25 == 0);
    const bool print_now = verbose && (i % 250 == 0);
    const bool last = (i == max_iters - 1);
    const bool plain_iter = !(check || print_now || last || csv);
    t = now_ms();
    w->project_lin_sys(i, use_graphs);  // ends with a stream sync (CG convergence flags)
    t_lin += now_ms() - t;
    t = now_ms();
    if (use_graphs && plain_iter) {
      HIP_CHECK(hipGraphLaunch(w->g_post, s));  // y, tau, cones, v += alpha (u - u_t)
      w->v_norm_fresh = true;
      t_cone += now_ms() - t;
    } else {
      w->enqueue_lin_sys_tail();
      w->enqueue_cones();
      if (!plain_iter)
        hipLaunchKernelGGL(k_rsk, dim3(w->vb(l)), dim3(kVecThreads), 0, s, w->rsk.p, w->v.p, w->u.p, w->ut.p, w->diag_r.p, l);
      t_cone += now_ms() - t;
      if (csv) w->populate_residuals(i);
      if (check) {
        w->populate_residuals(i);
        w->note_check_residuals();
        if ((info->status_val = w->has_converged(i)) != 0) {
          if (csv) w->log_csv_row(csv, i, now_ms() - t_start);
          break;
        }
        if (w->stgs.time_limit_secs > 0 && (now_ms() - t_start) > 1e3 * w->stgs.time_limit_secs) break;
      }
      if (print_now) {
        w->populate_residuals(i);
        std::printf("%6d|%9.2e|%9.2e|%9.2e|%9.2e|%9.2e|%9.2e\n", i, w->r.res_pri, w->r.res_dual, w->r.gap,
                    0.5 * (w->r.pobj + w->r.dobj), w->scale, (now_ms() - t_start) / 1e3);
        std::fflush(stdout);
      }
      if (w->stgs.adaptive_scale && check && i == w->r.last_iter) w->update_scale(i);
      w->enqueue_v_update();
      if (csv) w->log_csv_row(csv, i, now_ms() - t_start);
    }
    if (aa_now) {
      t = now_ms();
      w->aa_safeguard();
      t_acc += now_ms() - t;
    }
  }
  if (csv) std::fclose(csv);
  if (ScsHipWork::debug_pipe()) {
    std::fprintf(stderr, "[scs-hip] iterations %d, run-ahead stalls %d, CG steps of the last 8 solves:", i, w->pipe_stalls);
    for (int v : w->cg_hist) std::fprintf(stderr, " %d", v);
    std::fprintf(stderr, "\n");
  }
  w->finish_solve(sol, info, i, t_start, t_lin, t_cone, t_acc);
  if (verbose) {
    std::printf("------------------------------------------------------------------\n");
    std::printf("status:  %s\ntimings: total: %.2es = setup: %.2es + solve: %.2es\n\t lin-sys: %.2es, cones: %.2es, accel: %.2es\n",
                info->status, (info->setup_time + info->solve_time) / 1e3, info->setup_time / 1e3, info->solve_time / 1e3,
                t_lin / 1e3, t_cone / 1e3, t_acc / 1e3);
    std::printf("lin-sys: avg cg its: %.2f\n", info->iter > 0 ? (double)w->tot_cg_iters / (info->iter + 1) : 0.0);
    std::printf("------------------------------------------------------------------\n");
    // solution / certificate quality, the block the reference prints here (R:notebooks/scs_benchmarks.ipynb cells 2, 3)
    switch (info->status_val) {
      case SCS_SOLVED:
      case SCS_SOLVED_INACCURATE: {
        double sy = 0., ns = 0., ny = 0.;
        for (int j = 0; j < m; ++j) { sy += sol->s[j] * sol->y[j]; ns += sol->s[j] * sol->s[j]; ny += sol->y[j] * sol->y[j]; }
        std::printf("cones: dist(s, K) = %.2e, dist(y, K*) = %.2e\n", w->cone_dist(sol->s, 0), w->cone_dist(sol->y, 1));
        std::printf("comp slack: s'y/|s||y| = %.2e, gap: |x'Px+c'x+b'y| = %.2e\n", safediv_pos(sy, std::sqrt(ns) * std::sqrt(ny)), info->gap);
        std::printf("pri res: |Ax+s-b| = %.2e, dua res: |Px+A'y+c| = %.2e\n", info->res_pri, info->res_dual);
        break;
      }
      case SCS_INFEASIBLE:
      case SCS_INFEASIBLE_INACCURATE:
        std::printf("cone: dist(y, K*) = %.2e\n", w->cone_dist(sol->y, 1));
        std::printf("cert: |A'y| = %.2e\n      b'y = %.2f\n", info->res_infeas, -1.0);
        break;
      default:
        std::printf("cone: dist(s, K) = %.2e\n", w->cone_dist(sol->s, 0));
        std::printf("cert: |Ax+s| = %.2e\n      |Px| = %.2e\n      c'x = %.2f\n", info->res_unbdd_a, info->res_unbdd_p, -1.0);
        break;
    }
    std::printf("------------------------------------------------------------------\n");
    std::printf("objective = %.6f\n", info->pobj);
    std::printf("------------------------------------------------------------------\n");
    std::fflush(stdout);
  }
  return info->status_val;
}

