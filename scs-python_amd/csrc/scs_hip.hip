// scs_hip.hip — libscs_hip.so: device-resident SCS ADMM loop for MI355X (gfx950).
//
// Replaces, for the hot path only, the absent SCS C core behind the reference's
// glue: scs_init / scs_solve / scs_update / scs_finish (R:scs/scsobject.h:903,986,
// 1217,1240), i.e. scs_source/src/scs.c + linsys/gpu/indirect (R:meson.build:195,
// 303-304).  Not a port: the reference's GPU backend keeps only the CG mat-vecs on
// the device (cuSPARSE/cuBLAS) and round-trips rhs/solution over PCIe every
// iteration; here the whole iteration — KKT solve, cone
// projections, Anderson acceleration, residuals — stays in HBM, and only a few
// scalars per iteration plus the final x,y,s cross PCIe.
//
// Iteration, state v, R = diag(diag_r):
//   u_t = (R+Q)^{-1} R v   : PCG on (R_x + P + A' R_y^{-1} A), then tau from a quadratic
//   u   = Pi_{R^n x K* x R+}(2 u_t - v)
//   rsk = R (v + u - 2 u_t)   (only when residuals are needed)
//   v  += alpha (u - u_t)
#include <atomic>
#include <chrono>
#include <csignal>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "options.hpp"
#include "aa.hpp"
#include "common.hpp"
#include "cones.hpp"
#include "host_setup.hpp"
#include "normalize_dev.hpp"
#include "psd.hpp"
#ifdef SCS_HIP_LABS  // experiments that lost their measurement (options.hpp): compiled into libscs_hip_labs.so only
#include "cg_persist.hpp"
#include "minres.hpp"
#endif
#include "dense.hpp"
#include "setup_dev.hpp"
#include "setup_cs_dev.hpp"
#include "spmv.hpp"
#include "vec.hpp"
#ifdef SCS_HIP_LABS
#include "cg_k1dot.hpp"
#endif

#include "runtime.hpp"
#include "device_csr.hpp"
#include "work.hpp"
#include "io.hpp"
#include "setup.hpp"
#include "loop.hpp"
#include "batch.hpp"

// ================================================================ C ABI
extern "C" {

ScsWork *scs_init(const ScsData *d, const ScsCone *k, const ScsSettings *stgs) {
  try {
    set_last_error("");
    refresh_options();
    return init_impl(d, k, stgs);
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return nullptr;
  }
}

ScsWork *scs_hip_init_linsys(const ScsData *d, const ScsCone *k, const ScsSettings *stgs, int linsys) {
  try {
    set_last_error("");
    refresh_options();
    return init_impl(d, k, stgs, linsys);
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return nullptr;
  }
}
int scs_hip_linsys_kind(const ScsWork *w) { return w ? (w->dense() ? 2 : 1) : 0; }

scs_int scs_solve(ScsWork *w, ScsSolution *sol, ScsInfo *info, scs_int warm_start) {
  if (!w || !sol || !info) return SCS_FAILED;
  try {
    set_last_error("");
    refresh_options();
    const double scale_entry = w->scale;
    try {
      return solve_impl(w, sol, info, warm_start);
    } catch (const SpinTimeout &) {
      // not a failure of the problem: restart without the spinning kernels (the caller's sol is untouched until a solve finishes)
      ++g_spin_fallbacks;
      {
        std::lock_guard<std::mutex> lock(w->mtx);
        HIP_CHECK(hipSetDevice(w->device));
        w->spin_fallback(scale_entry);
      }
      return solve_impl(w, sol, info, warm_start);
    }
  } catch (const std::exception &e) {
    set_last_error(e.what());
    info->status_val = SCS_FAILED;
    std::snprintf(info->status, sizeof(info->status), "failure");
    fill_nan(sol->x, w->n);
    fill_nan(sol->y, w->m);
    fill_nan(sol->s, w->m);
    return SCS_FAILED;
  }
}

// Grouped solve of `count` workspaces (include/scs_hip.h): members that can share launches — same shape, CSR-stream
// layouts, one-launch cone kernels (GroupSolve::member_ok / same_shape) — advance in lock step through the grouped
// kernels of batch.hpp; the others are solved by scs_solve's own loop.  Every info[i] / sol[i] is filled exactly as
// scs_solve(w[i], sol[i], info[i], warm_start) would (iterates are bit-identical).
// A shape class is one group (up to SCS_HIP_GROUP_MAX members, default 1024).  Cutting it into several groups that run
// concurrently — one host thread and one stream each, SCS_HIP_GROUP_LANES > 1 — was built and measured (512 config-5
// problems, solve phase): 1 x 512: 4.0 s, 4 x 128: 4.2 s, 8 x 64: 6.4 s, 8 lanes of 32: 10.7 s — launches and
// synchronisations issued from several host threads contend inside the HIP runtime, as the one-stream-per-problem
// mode showed before (profiles/r02_batch_queues.txt) => off by default.
static scs_int solve_one_group(ScsWork **works, ScsSolution **sols, ScsInfo **infos, const std::vector<int> &idx, scs_int warm_start,
                               std::string &err) {
  if (idx.size() == 1) {
    const int i = idx[0];
    const scs_int st = scs_solve(works[i], sols[i], infos[i], warm_start);
    if (st == SCS_FAILED) { err = scs_hip_last_error(); return -1; }
    return 0;
  }
  scs_int rc = 0;
  std::vector<std::unique_lock<std::mutex>> locks;
  std::vector<hipStream_t> saved;
  GroupSolve gs;
  //  every member starts out "unfinished": the catch block below marks exactly those the solve did not finish, also when
  // the exception comes before the members' begin_solve cleared their infos (caller memory); member mutexes are taken in ADDRESS
  // order, so two concurrent batches that share workspaces in different orders cannot deadlock (the Python glue sorts too)
  for (int j : idx) {
    std::memset(infos[j], 0, sizeof(ScsInfo));
    infos[j]->status_val = SCS_UNFINISHED;
  }
  try {
    {
      std::vector<int> order(idx);
      std::sort(order.begin(), order.end(), [&](int a, int b) { return std::less<ScsHipWork *>()(works[a], works[b]); });
      for (int j : order) locks.emplace_back(works[j]->mtx);
    }
    HIP_CHECK(hipSetDevice(works[idx[0]]->device));
    gs.s = works[idx[0]]->stream;
    for (int j : idx) {
      gs.W.push_back(works[j]); gs.sols.push_back(sols[j]); gs.infos.push_back(infos[j]);
      saved.push_back(works[j]->stream);
      works[j]->stream = gs.s;  // every member's kernels go to the group's stream for the duration of the solve
      works[j]->aa.stream = gs.s;
    }
    gs.build();
    gs.run(warm_start);
  } catch (const std::exception &e) {
    err = e.what();
    rc = -1;
    (void)hipStreamSynchronize(gs.s);
    for (int j : idx)
      if (infos[j]->status[0] == 0) {  // (finish_solve writes the status string: empty = this member never got there)
        infos[j]->status_val = SCS_FAILED;
        std::snprintf(infos[j]->status, sizeof(infos[j]->status), "failure");
        fill_nan(sols[j]->x, works[j]->n);
        fill_nan(sols[j]->y, works[j]->m);
        fill_nan(sols[j]->s, works[j]->m);
      }
  }
  for (size_t k = 0; k < saved.size(); ++k) {
    works[idx[k]]->stream = saved[k];
    works[idx[k]]->aa.stream = saved[k];
  }
  return rc;
}

scs_int scs_hip_solve_batch(ScsWork **works, ScsSolution **sols, ScsInfo **infos, scs_int count, scs_int warm_start) {
  if (!works || !sols || !infos || count < 0) return -1;
  set_last_error("");
  InterruptListener ctrlc;  // for the whole call: members solved one after the other all see the same Ctrl-C
  for (int i = 0; i < count; ++i) {
    if (!works[i] || !sols[i] || !infos[i]) { set_last_error("scs_hip_solve_batch: null entry"); return -1; }
    for (int j = 0; j < i; ++j)
      if (works[j] == works[i]) { set_last_error("scs_hip_solve_batch: a workspace appears twice"); return -1; }
  }
  refresh_options();
  const int group_max = opts().group_max, lanes = opts().group_lanes, group_min = opts().group_min;  // (labs knobs: several concurrently driven groups lost)
  // shape classes, then groups
  std::vector<std::vector<int>> jobs;
  std::vector<char> taken((size_t)count, 0);
  for (int i = 0; i < count; ++i) {
    if (taken[(size_t)i]) continue;
    taken[(size_t)i] = 1;
    std::vector<int> cls{i};
    if (group_max > 1 && GroupSolve::member_ok(works[i]))
      for (int j = i + 1; j < count; ++j)
        if (!taken[(size_t)j] && GroupSolve::member_ok(works[j]) && GroupSolve::same_shape(works[i], works[j])) {
          taken[(size_t)j] = 1;
          cls.push_back(j);
        }
    const int S = (int)cls.size();
    int k = std::max((S + group_max - 1) / group_max, lanes > 1 ? std::min(lanes, S / group_min) : 1);
    k = std::max(1, std::min(k, S));
    for (int part = 0; part < k; ++part) {  // contiguous, near-equal parts
      const int lo = (int)((long)S * part / k), hi = (int)((long)S * (part + 1) / k);
      jobs.emplace_back(cls.begin() + lo, cls.begin() + hi);
    }
  }
  std::stable_sort(jobs.begin(), jobs.end(), [](const std::vector<int> &a, const std::vector<int> &b) { return a.size() > b.size(); });
  const int nthreads = std::max(1, std::min(lanes, (int)jobs.size()));
  std::atomic<size_t> next{0};
  std::atomic<int> rc_all{0};
  std::mutex err_mtx;
  std::string first_err;
  const int dflt_dev = current_device();
  auto worker = [&]() {
    (void)hipSetDevice(dflt_dev);
    while (true) {
      const size_t j = next.fetch_add(1);
      if (j >= jobs.size()) break;
      std::string err;
      if (solve_one_group(works, sols, infos, jobs[j], warm_start, err) != 0) {
        rc_all.store(-1);
        std::lock_guard<std::mutex> g(err_mtx);
        if (first_err.empty()) first_err = err;
      }
    }
  };
  if (nthreads == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
  }
  if (rc_all.load() != 0) set_last_error(first_err);
  return rc_all.load();
}

scs_int scs_update(ScsWork *w, scs_float *b, scs_float *c) {
  if (!w) return -1;
  try {
    std::lock_guard<std::mutex> lock(w->mtx);
    HIP_CHECK(hipSetDevice(w->device));
    const int n = w->n, m = w->m;
    if (b) w->b_orig.assign(b, b + m);
    if (c) w->c_orig.assign(c, c + n);
    w->nm_b_orig = w->nm_c_orig = 0;
    for (double x : w->b_orig) w->nm_b_orig = std::max(w->nm_b_orig, std::fabs(x));
    for (double x : w->c_orig) w->nm_c_orig = std::max(w->nm_c_orig, std::fabs(x));
    std::vector<double> bn(w->b_orig), cn(w->c_orig);
    if (w->normalized) normalize_b_c(w->scal, bn.data(), m, cn.data(), n);
    std::vector<double> hh(w->l, 0.0);
    std::copy(cn.begin(), cn.end(), hh.begin());
    std::copy(bn.begin(), bn.end(), hh.begin() + n);
    w->h.upload(hh.data(), w->l, w->stream);
    if (w->normalized) {  // sigma changed: refresh the un-normalisation factors
      std::vector<double> di(m), ei(n);
      for (int i = 0; i < m; ++i) di[i] = 1.0 / (w->scal.D[i] * w->scal.sigma);
      for (int i = 0; i < n; ++i) ei[i] = 1.0 / (w->scal.E[i] * w->scal.sigma);
      w->Dinv.upload(di.data(), m, w->stream);
      w->Einv.upload(ei.data(), n, w->stream);
    }
    HIP_CHECK(hipStreamSynchronize(w->stream));
    if (w->setup_pending) w->finish_pending_setup();  // (a workspace that has not solved yet: R, the preconditioner or G^{-1}, and g in one go)
    else w->update_work_cache();
    HIP_CHECK(hipStreamSynchronize(w->stream));
    return 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

void scs_finish(ScsWork *w) {
  if (!w) return;
  try {
    (void)hipSetDevice(w->device);
    if (w->stream) (void)hipStreamSynchronize(w->stream);
  } catch (...) {
  }
  delete w;
}

void scs_set_default_settings(ScsSettings *s) {
  // defaults: R:README.md:98-104 (AA); R:test/test_warm_start_consistency.py:228-241 (scale, rho_x, alpha);
  // banner R:notebooks/scs_benchmarks.ipynb cell 2 (eps, max_iters, normalize, adaptive_scale)
  s->normalize = 1;
  s->scale = 0.1;
  s->adaptive_scale = 1;
  s->rho_x = 1e-6;
  s->max_iters = 100000;
  s->eps_abs = 1e-4;
  s->eps_rel = 1e-4;
  s->eps_infeas = 1e-7;
  s->alpha = 1.5;
  s->time_limit_secs = 0.;
  s->verbose = 1;
  s->warm_start = 0;
  s->acceleration_lookback = 10;
  s->acceleration_interval = 10;
  s->acceleration_type_1 = 1;
  s->acceleration_regularization = 1e-8;
  s->acceleration_relaxation = 1.0;
  s->write_data_filename = nullptr;
  s->log_csv_filename = nullptr;
}

const char *scs_version(void) { return "3.2.11"; }
size_t scs_sizeof_int(void) { return sizeof(scs_int); }
size_t scs_sizeof_float(void) { return sizeof(scs_float); }

int scs_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int scs_hip_set_device(int dev) {
  int n = scs_hip_device_count();
  if (dev < 0 || dev >= n) return -1;
  g_default_device.store(dev);
  return 0;
}
int scs_hip_set_thread_device(int dev) {
  if (dev < 0) { t_device = -1; return 0; }
  if (dev >= scs_hip_device_count()) return -1;
  t_device = dev;
  return 0;
}
int scs_hip_labs_build(void) { return kLabsBuild ? 1 : 0; }
int scs_hip_mem_info(size_t *free_bytes, size_t *total_bytes) {
  if (scs_hip_device_count() <= 0) return -1;
  size_t f = 0, t = 0;
  if (hipSetDevice(current_device()) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return -1;
  if (free_bytes) *free_bytes = f + DevPool::inst().held_bytes();
  if (total_bytes) *total_bytes = t;
  return 0;
}
const char *scs_hip_last_error(void) { return g_last_error.c_str(); }

/* reps back-to-back launches of K1, of K2 and (QPs) of K3 on the solver's own stream and resident data, one HIP event
 * pair around each batch (event overhead amortised); out = {K1 avg ms, K2 avg ms, K3 avg ms inside the CG step's sequence, K3 back to back}
 * (K3: 0 without P) */
int scs_hip_time_matvec(ScsWork *w, int reps, double *out) {
  if (!w || !out || reps <= 0) return -1;
  try {
    std::lock_guard<std::mutex> lock(w->mtx);
    HIP_CHECK(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    w->finish_pending_setup();  // (R lives in the products)
    // the products exactly as the CG step of this workspace launches them (k1dot: cg_k1dot.hpp)
    for (int i = 0; i < 2; ++i) {
#ifdef SCS_HIP_LABS
      if (w->k1dot) { w->matvec_k1dot(w->cg_p.p, nullptr, nullptr); continue; }
#endif
      w->matvec(w->cg_p.p, nullptr);
    }
    HIP_CHECK(hipEventRecord(w->ev[0], s));
    for (int i = 0; i < reps; ++i) {
#ifdef SCS_HIP_LABS
      if (w->k1dot) { launch_spmv(w->Ar.view(), w->cg_p.p, EpiDivRDot{w->tmp_m.p, w->rdy(), w->part_k1.p}, nullptr, s); continue; }
#endif
      launch_spmv(w->Ar.view(), w->cg_p.p, EpiDivR{w->tmp_m.p, w->rdy()}, nullptr, s);
    }
    HIP_CHECK(hipEventRecord(w->ev[1], s));
    for (int i = 0; i < reps; ++i) {
#ifdef SCS_HIP_LABS
      if (w->k1dot) { launch_spmv(w->At.view(), w->tmp_m.p, EpiAtRaw{w->cg_Gp.p, w->gp2()}, nullptr, s); continue; }
#endif
      launch_spmv(w->At.view(), w->tmp_m.p, EpiGp{w->cg_Gp.p, w->cg_p.p, w->rdx(), w->has_P ? 1 : 0, w->part.p, w->gp2()}, nullptr, s);
    }
    HIP_CHECK(hipEventRecord(w->ev[2], s));
    HIP_CHECK(hipEventSynchronize(w->ev[2]));
    float a = 0, b = 0, c = 0;
    HIP_CHECK(hipEventElapsedTime(&a, w->ev[0], w->ev[1]));
    HIP_CHECK(hipEventElapsedTime(&b, w->ev[1], w->ev[2]));
    float c2 = 0;
    if (w->has_P) {  // K3: Gp = P p (csrc/spmv*.hpp on the full symmetric CSR of P, epilogue EpiStore)
      // (i) back to back: Pf alone (224 MB at the bench's target_qp) stays in the 256 MB Infinity Cache between launches — flattering;
      HIP_CHECK(hipEventRecord(w->ev[0], s));
      for (int i = 0; i < reps; ++i) launch_spmv(w->Pf.view(), w->cg_p.p, EpiStore{w->cg_Gp.p, 0}, nullptr, s);
      HIP_CHECK(hipEventRecord(w->ev[1], s));
      // (ii) as the CG step runs it, between K1 and K2, which stream 240 MB each: (K1, K3, K2) x reps minus (K1, K2) x reps
      for (int i = 0; i < reps; ++i) w->matvec(w->cg_p.p, nullptr);
      HIP_CHECK(hipEventRecord(w->ev[2], s));
      HIP_CHECK(hipEventSynchronize(w->ev[2]));
      HIP_CHECK(hipEventElapsedTime(&c, w->ev[0], w->ev[1]));
      float t_all = 0, t_12 = 0;
      HIP_CHECK(hipEventElapsedTime(&t_all, w->ev[1], w->ev[2]));
      HIP_CHECK(hipEventRecord(w->ev[0], s));
      for (int i = 0; i < reps; ++i) {
        launch_spmv(w->Ar.view(), w->cg_p.p, EpiDivR{w->tmp_m.p, w->rdy()}, nullptr, s);
        launch_spmv(w->At.view(), w->tmp_m.p, EpiGp{w->cg_Gp.p, w->cg_p.p, w->rdx(), 1, w->part.p, w->gp2()}, nullptr, s);
      }
      HIP_CHECK(hipEventRecord(w->ev[1], s));
      HIP_CHECK(hipEventSynchronize(w->ev[1]));
      HIP_CHECK(hipEventElapsedTime(&t_12, w->ev[0], w->ev[1]));
      c2 = t_all - t_12;
    }
    out[0] = a / reps;
    out[1] = b / reps;
    out[2] = c2 / reps;
    out[3] = c / reps;
    return 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

int scs_hip_solution_to_device(ScsWork *w, scs_float *x_dev, scs_float *y_dev, scs_float *s_dev) {
  if (!w) return -1;
  try {
    std::lock_guard<std::mutex> lock(w->mtx);
    if (!w->sol_on_device) throw std::runtime_error("no solution yet: call scs_solve first");
    HIP_CHECK(hipSetDevice(w->device));
    if (x_dev) HIP_CHECK(hipMemcpyAsync(x_dev, w->solx.p, sizeof(double) * w->n, hipMemcpyDeviceToDevice, w->stream));
    if (y_dev) HIP_CHECK(hipMemcpyAsync(y_dev, w->soly.p, sizeof(double) * w->m, hipMemcpyDeviceToDevice, w->stream));
    if (s_dev) HIP_CHECK(hipMemcpyAsync(s_dev, w->sols.p, sizeof(double) * w->m, hipMemcpyDeviceToDevice, w->stream));
    HIP_CHECK(hipStreamSynchronize(w->stream));
    return 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

void scs_hip_set_mark(ScsWork *w, int iter) {
  if (w) w->mark_iter = iter;
}
void scs_hip_get_mark(const ScsWork *w, double *out) {
  if (!w || !out) return;
  out[0] = w->mark_ms; out[1] = (double)w->mark_cg; out[2] = (double)w->mark_aa_calls; out[3] = (double)w->mark_aa_accept;
}

/* bench.py --workload config4_psd: average duration of one batched PSD projection (K9, all s-cones of the problem) on the
 * solver's own stream and resident state — the current dual iterate is copied to a scratch vector and projected `reps`
 * times (warm-started eigenvectors, as inside the ADMM loop); the copies are timed separately and subtracted.
 * out[4] = {ms per projection, number of matrices, largest order, flops of a LAPACK-style eigensolve of them all
 * (16/3 n^3 + 2 n^3 per matrix)}.  Returns 0 on success, 1 when the problem has no PSD cone. */
long scs_hip_spin_fallbacks(void) { return g_spin_fallbacks.load(); }

void scs_hip_trim_pool(void) { DevPool::inst().trim(); }

int scs_hip_psd_refine_stats(ScsWork *w, double *out, int cap) {
  if (!w || !out || cap < 0) return -1;
  try {
    std::lock_guard<std::mutex> lock(w->mtx);
    HIP_CHECK(hipSetDevice(w->device));
    HIP_CHECK(hipStreamSynchronize(w->stream));
    const int cnt = std::min(cap, w->n_psd_big);
    for (int c = 0; c < cnt; ++c) {
      double st[kPsdStateDoubles];
      const long at = w->psd_woff_h[(size_t)c] + psd_scratch_doubles(w->psd_order_h[(size_t)c]) - kPsdStateDoubles;
      HIP_CHECK(hipMemcpy(st, w->psd_scratch.p + at, sizeof st, hipMemcpyDeviceToHost));
      out[8 * c + 0] = st[9];
      out[8 * c + 1] = st[10];
      out[8 * c + 2] = st[8];
      out[8 * c + 3] = st[11];
      out[8 * c + 4] = st[7];
      out[8 * c + 5] = st[12];
      out[8 * c + 6] = st[13];
      out[8 * c + 7] = st[14];
    }
    return cnt;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

int scs_hip_time_psd(ScsWork *w, int reps, double *out) {
  if (!w || !out || reps <= 0) return -1;
  try {
    std::lock_guard<std::mutex> lock(w->mtx);
    HIP_CHECK(hipSetDevice(w->device));
    if (w->n_psd <= 0) return 1;
    hipStream_t s = w->stream;
    const size_t bytes = sizeof(double) * w->m;
    auto copy = [&] { HIP_CHECK(hipMemcpyAsync(w->tmp_m.p, w->u.p + w->n, bytes, hipMemcpyDeviceToDevice, s)); };
    for (int i = 0; i < 2; ++i) { copy(); w->launch_psd(w->tmp_m.p, w->psd_off.p, w->psd_order.p, w->psd_woff.p, w->n_psd, w->n_psd_big); }
    HIP_CHECK(hipEventRecord(w->ev[0], s));
    for (int i = 0; i < reps; ++i) copy();
    HIP_CHECK(hipEventRecord(w->ev[1], s));
    for (int i = 0; i < reps; ++i) { copy(); w->launch_psd(w->tmp_m.p, w->psd_off.p, w->psd_order.p, w->psd_woff.p, w->n_psd, w->n_psd_big); }
    HIP_CHECK(hipEventRecord(w->ev[2], s));
    HIP_CHECK(hipEventSynchronize(w->ev[2]));
    float a = 0, b = 0;
    HIP_CHECK(hipEventElapsedTime(&a, w->ev[0], w->ev[1]));
    HIP_CHECK(hipEventElapsedTime(&b, w->ev[1], w->ev[2]));
    double flops = 0.;
    int mx = 0;
    for (int sd : w->cone.s) { flops += (16. / 3. + 2.) * (double)sd * sd * sd; mx = std::max(mx, sd); }
    out[0] = (b - a) / reps;
    out[1] = (double)w->cone.s.size();
    out[2] = (double)mx;
    out[3] = flops;
    return 0;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return -1;
  }
}

void scs_hip_set_profiling(ScsWork *w, int on) {
  if (w) w->profile = on != 0;
}
void scs_hip_kernel_times(const ScsWork *w, double *out) {
  if (!w || !out) return;
  out[0] = w->prof_ms[0]; out[1] = (double)w->prof_n[0];
  out[2] = w->prof_ms[1]; out[3] = (double)w->prof_n[1];
  out[4] = (double)w->At.nnz; out[5] = (double)w->Ar.nwg(); out[6] = (double)w->At.nwg();
  out[7] = w->has_P ? (double)w->Pf.nnz : 0.0;
  out[8] = w->prof_cone_ms; out[9] = (double)w->prof_cone_n;
  out[10] = w->prof_ms[2]; out[11] = (double)w->prof_n[2];
}

#include "lab_entries.hpp"

}  // extern "C"

