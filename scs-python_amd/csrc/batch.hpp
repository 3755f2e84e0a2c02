// batch.hpp — grouped solve: G equally shaped, independent cone programs advance through the ADMM loop in lock
// step, ONE launch per kernel of the iteration for the whole group (BASELINE.json configs[4]).
//
// The reference's notion of a batch is "independent SCS instances may run concurrently"
// (R:test/test_thread_safety.py:78-93): one solver object, one stream each.  On this GPU that is bound by the
// command queues, not by the kernels: a small problem's iteration is ~35 dependent launches of ~5 us, a hardware
// queue retires them one after the other, and four queues give 4 x 5.4 k iterations/s however many streams are
// open (profiles/r02_batch_queues.txt).  So the group shares the launches instead: blockIdx.y selects the problem,
// blockIdx.x is what the one-problem launch would have used, and the kernel's arguments come from a per-problem
// record in HBM.
//
// No kernel is written twice.  Every kernel of the iteration is a __device__ body `d_X(args...)` with two
// entry points: the one-problem kernel `k_X(args...)` and the generic `k_grouped<d_X, threads, Args...>`, which
// loads the argument record of problem list[blockIdx.y] (scalar loads: the address is uniform) and calls the same
// body.  Same code, same block decomposition, same reduction partials => every problem's iterates, iteration and
// CG-step counts are bit-identical to a solve of its own (tests/test_group_gpu.py).
//
// Host control (GroupSolve::run) is the control of solve_impl applied to every member: the members share the
// iteration index, so the Anderson steps (every `acceleration_interval`) and the convergence checks (every 25th)
// fall together; what differs per member — CG steps of a linear solve, Anderson history length, accept / reject /
// safeguard, adaptive-scale updates, the iteration it converges in — is handled with sub-lists of the group (a
// device array of member indices per launch) and one read-back of all members' flags / residual records per
// decision point.  One host synchronisation per iteration (after the CG chunk), one more on check iterations.
#pragma once
#include <algorithm>
#include <numeric>

namespace scship {

// ---- argument records --------------------------------------------------------------------------------------
template <class... T> struct Pack;
template <> struct Pack<> {};
template <class H, class... T> struct Pack<H, T...> {
  H head;
  Pack<T...> tail;
};
inline Pack<> make_pack() { return {}; }
template <class H, class... T> inline Pack<H, T...> make_pack(H h, T... t) {
  Pack<H, T...> p;
  p.head = h;
  p.tail = make_pack(t...);
  return p;
}
template <class F, class... D> __device__ __forceinline__ void pack_call(F f, const Pack<> &, D... d) { f(d...); }
template <class F, class H, class... T, class... D>
__device__ __forceinline__ void pack_call(F f, const Pack<H, T...> &p, D... d) {
  pack_call(f, p.tail, d..., p.head);
}

template <auto Fn, int Threads, class... A>
__global__ __launch_bounds__(Threads) void k_grouped(const Pack<A...> *tab, const int *list) {
  pack_call(Fn, tab[list[blockIdx.y]]);
}

// One kernel of the group: the members' argument records (pinned host copy + HBM copy) and the launch.
// The host copy is pinned because uploads are asynchronous: a record may only be rewritten after the host has
// synchronised with the stream since the last upload() (GroupSolve does, once per iteration).
template <auto Fn, int Threads, class... A>
struct GTable {
  using Rec = Pack<A...>;
  Rec *host = nullptr;
  size_t count = 0;
  DevBuf<Rec> dev;
  int gx = 1;        // blockIdx.x extent (what the one-problem launch uses; the largest over the members)
  size_t lds = 0;    // dynamic LDS bytes
  bool used = false;
  GTable() = default;
  GTable(const GTable &) = delete;
  GTable &operator=(const GTable &) = delete;
  ~GTable() { if (host) (void)hipHostFree(host); }
  void resize(size_t G) {
    if (host) (void)hipHostFree(host);
    host = nullptr;
    HIP_CHECK(hipHostMalloc((void **)&host, sizeof(Rec) * G));
    std::memset((void *)host, 0, sizeof(Rec) * G);
    count = G;
    used = true;
  }
  void set(int g, A... a) { host[(size_t)g] = make_pack(a...); }
  void upload(hipStream_t s) {
    if (!used) return;
    if (dev.n != count) dev.alloc(count);
    HIP_CHECK(hipMemcpyAsync(dev.p, host, sizeof(Rec) * count, hipMemcpyHostToDevice, s));
  }
  void launch(const int *list, int cnt, hipStream_t s) const {
    if (!used || cnt <= 0 || gx <= 0) return;
    hipLaunchKernelGGL((k_grouped<Fn, Threads, A...>), dim3((unsigned)gx, (unsigned)cnt), dim3(Threads), lds, s,
                       (const Rec *)dev.p, list);
  }
};
template <auto Fn, int Threads, class... A> GTable<Fn, Threads, A...> gtable_of(void (*)(A...));
#define SCS_GTABLE(threads, ...) decltype(gtable_of<__VA_ARGS__, threads>(__VA_ARGS__))

// table bodies take their workgroup index from the launch
__device__ __forceinline__ void d_soc_wave_blk(double *x, const int *__restrict__ off, const int *__restrict__ dim, int ncones, int G,
                                               const int *stall) {
  d_proj_soc_wave(x, off, dim, ncones, G, stall, (int)blockIdx.x);
}
__device__ __forceinline__ void d_psd_small4_blk(double *x, PsdBatch B, double *scratch, int allow_warm, const int *stall,
                                                 const double *tol2) {
  d_proj_psd_small4(x, B, scratch, allow_warm, stall, tol2, (int)blockIdx.x);
}

// ---- small bodies only the grouped path needs (the one-problem path uses hipMemset / hipMemcpy for these) ----
__device__ __forceinline__ void d_copy_f64(const double *__restrict__ src, double *dst, long n) {
  for (long i = (long)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (long)gridDim.x * kVecThreads) dst[i] = src[i];
}
__device__ __forceinline__ void d_copy_i32(const int *__restrict__ src, int *dst, int n) {
  for (int i = threadIdx.x; i < n; i += kVecThreads) dst[i] = src[i];
}
__device__ __forceinline__ void d_fill_f64(double *dst, double v, int n) {
  for (int i = threadIdx.x; i < n; i += kVecThreads) dst[i] = v;
}
__device__ __forceinline__ void d_fill_i32(int *dst, int v, int n) {
  for (int i = threadIdx.x; i < n; i += kVecThreads) dst[i] = v;
}
// residual record of a member: the 32 reduced scalars of the two residual products, then u_tau, rsk_tau
constexpr int kResRec = 34;
__device__ __forceinline__ void d_gather_res(const double *__restrict__ out, const double *utau, const double *rtau, double *dst) {
  const int t = threadIdx.x;
  if (t < 32) dst[t] = out[t];
  else if (t == 32) dst[32] = *utau;
  else if (t == 33) dst[33] = *rtau;
}

// ---- the group ---------------------------------------------------------------------------------------------
struct GroupSolve {
  std::vector<ScsHipWork *> W;
  std::vector<ScsSolution *> sols;
  std::vector<ScsInfo *> infos;
  int G = 0, n = 0, m = 0;
  long l = 0;
  hipStream_t s = nullptr;
  bool has_P = false;
  std::vector<char> mr_allowed_saved;

  // tables (one per kernel of the path; names follow the kernels)
  SCS_GTABLE(kVecThreads, d_sumsq) t_sumsq;
  SCS_GTABLE(kVecThreads, d_prep) t_prep;
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiY>) t_spmv_y;
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiStore>) t_spmv_pws, t_spmv_p, t_res_px, t_spmv_ax;
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiR0>) t_spmv_r0;
  SCS_GTABLE(kVecThreads, d_fin_head) t_fin_head;
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiDivR>) t_spmv_a;
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiGp>) t_spmv_at;
  SCS_GTABLE(kVecThreads, d_cg_update) t_cg_update[2];  // [0] ADMM (x = ut, y carried along), [1] cold KKT solve (x = ws)
  SCS_GTABLE(kVecThreads, d_cg_dir) t_cg_dir[2];
  SCS_GTABLE(kVecThreads, d_tau_dots) t_tau_dots;
  SCS_GTABLE(kVecThreads, d_cone_pre) t_cone_pre;
  SCS_GTABLE(kBoxThreads, d_proj_box) t_box;
  SCS_GTABLE(kConeThreads, d_soc_wave_blk) t_soc;
  SCS_GTABLE(kPsdSmallThreads, d_psd_small4_blk) t_psd;
  SCS_GTABLE(kPsdSmallThreads, d_proj_soc_psd_small) t_soc_psd;  // both in one launch (members with short SOCs and small PSD matrices)
  SCS_GTABLE(kConeThreads, d_proj_exp) t_exp_p, t_exp_d;
  SCS_GTABLE(kConeThreads, d_proj_pow_dual) t_pow;
  SCS_GTABLE(kVecThreads, d_v_update) t_v_update;
  SCS_GTABLE(kVecThreads, d_rsk) t_rsk;
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiResPri>) t_res_pri;
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiResDual>) t_res_dual;
  SCS_GTABLE(kVecThreads, d_fin_multi) t_fin_multi_p, t_fin_multi_d;
  SCS_GTABLE(kVecThreads, d_gather_res) t_gather_res;
  SCS_GTABLE(kVecThreads, d_copy_i32) t_gather_fl;
  // dense direct linsys (dense.hpp): rhs = R_x v_x - A' v_y; u~_x = G^{-1} rhs; u~_y = v_y + R_y^{-1} A u~_x — and the cold KKT solve of a scale update
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiDenseRhs>) t_dense_rhs;
  SCS_GTABLE(kDenseThreads, d_dense_gemv) t_dense_gemv, t_dense_gemv_kkt;   // SCS_HIP_DENSE_GEMV=full
  SCS_GTABLE(kDenseThreads, d_dense_symv_tiles) t_symv_tiles;
  SCS_GTABLE(kDenseThreads, d_dense_symv_sum) t_symv_sum, t_symv_sum_kkt;
  void go_dense_gemv(const int *list, int count, bool kkt) {  // x = G^{-1} cg_b into ut (iteration) or ws (cold KKT solve)
    if (ScsHipWork::dense_full_gemv()) { go(kkt ? t_dense_gemv_kkt : t_dense_gemv, list, count); return; }
    go(t_symv_tiles, list, count);
    go(kkt ? t_symv_sum_kkt : t_symv_sum, list, count);
  }
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiY>) t_dense_y;
  bool dense = false;
  int dn_NP = 0;
  DevBuf<DenseMat> dmat_d;
  DevBuf<DenseSrc> dsrc_d;
  // adaptive-scale update of a sub-list
  SCS_GTABLE(kVecThreads, d_set_diag_r) t_set_diag_r;
  SCS_GTABLE(kVecThreads, d_precond) t_precond;
  SCS_GTABLE(kVecThreads, d_g_rhs) t_g_rhs;
  SCS_GTABLE(kVecThreads, d_kkt_prep) t_kkt_prep;
  SCS_GTABLE(kSpmvThreads, d_spmv_stream<EpiRhs>) t_spmv_rhs;
  SCS_GTABLE(kVecThreads, d_fill_f64) t_zero_part;
  SCS_GTABLE(kVecThreads, d_fin_tol) t_fin_tol;
  SCS_GTABLE(kVecThreads, d_cg_init) t_cg_init;
  SCS_GTABLE(kVecThreads, d_fin_cg_init) t_fin_cg_init;
  SCS_GTABLE(kVecThreads, d_fill_i32) t_zero_iters;
  SCS_GTABLE(kVecThreads, d_kkt_y) t_kkt_y;
  SCS_GTABLE(kVecThreads, d_copy_f64) t_copy_g, t_gather_aa;
  SCS_GTABLE(kVecThreads, d_gg) t_gg;
  SCS_GTABLE(kVecThreads, d_fin_store_sum) t_fin_gg;
  SCS_GTABLE(kVecThreads, d_v_rescale) t_v_rescale;
  // Anderson acceleration
  SCS_GTABLE(kVecThreads, d_aa_seed) t_aa_seed;
  SCS_GTABLE(kVecThreads, d_aa_update) t_aa_update;
  static constexpr int kMaxTsqrLevels = 4;  // (the reduction tree is 1024 -> 16 -> 1 wavefronts: at most three launches)
  SCS_GTABLE(64, d_aa_tsqr) t_aa_tsqr[kMaxTsqrLevels];
  int n_tsqr = 0, tsqr_fast[kMaxTsqrLevels] = {0, 0, 0, 0};  // default histories: the register-tile kernel (aa.hpp aa_tsqr_fast_kind)
  SCS_GTABLE(64 * kAaFastWaves, d_aa_tsqr_fast<21, 10>) t_aa_tsqr_f21[kMaxTsqrLevels];
  SCS_GTABLE(64 * kAaFastWaves, d_aa_tsqr_fast<11, 10>) t_aa_tsqr_f11[kMaxTsqrLevels];
  SCS_GTABLE(64, d_aa_solve) t_aa_solve;
  SCS_GTABLE(kVecThreads, d_aa_apply) t_aa_apply;
  SCS_GTABLE(kVecThreads, d_aa_diffsq) t_aa_diffsq;
  SCS_GTABLE(kVecThreads, d_fin_safeguard) t_fin_safe;
  SCS_GTABLE(kVecThreads, d_aa_restore) t_aa_restore;

  // group-owned device / pinned memory
  DevBuf<double> params_d, res_d, aa_res_d;
  DevBuf<int> flags_d, lists_d, active_buf;
  static constexpr int kParamRing = 32;  // pinned staging slots of the per-iteration parameter blocks (dense mode runs without a sync per iteration)
  int iters_since_sync = 0;
  std::vector<double> params_last;  // the block as last uploaded
  bool params_dirty = true;
  double *params_h = nullptr, *res_h = nullptr, *aa_res_h = nullptr;
  int *flags_h = nullptr, *lists_h = nullptr, *active_h = nullptr;  // (all pinned: every copy here is asynchronous)
  static constexpr int kListSlots = 64;
  int list_slot = 0, lists_since_sync = 0;
  bool soc_psd_fused = false;     // short SOCs and small PSD matrices of a member in one launch
  const int *active_d = nullptr;  // device copy of `active`
  double t_finish = 0.;           // host time inside finish_solve (SCS_HIP_DEBUG=group)
  std::vector<int> active;

  // per-member host state of this solve
  std::vector<double> cg_res_min;
  std::vector<int> aa_mode, aa_len;
  int mem = 0, interval = 1;
  double t_start = 0, t_lin = 0, t_cone = 0, t_acc = 0;
  long launches = 0;  // grouped launches issued (diagnostics: SCS_HIP_DEBUG=group)
  int syncs = 0, lockstep_iters = 0;

  ~GroupSolve() {
    if (params_h) (void)hipHostFree(params_h);
    if (res_h) (void)hipHostFree(res_h);
    if (aa_res_h) (void)hipHostFree(aa_res_h);
    if (flags_h) (void)hipHostFree(flags_h);
    if (lists_h) (void)hipHostFree(lists_h);
    if (active_h) (void)hipHostFree(active_h);
  }

  // Can these workspaces advance as one group?  Same dimensions and cone structure (every launch geometry follows
  // from them), the plain CSR-stream layouts, cone kernels that are one launch each, the same Anderson schedule.
  static bool member_ok(const ScsHipWork *w) {
    if (w->At.cs.ok || w->Ar.cs.ok || w->At.has_slab || w->Ar.has_slab) return false;
    if (w->has_P && (w->Pf.cs.ok || w->Pf.has_slab)) return false;
    if (w->persist_wgs > 0 || !w->log_csv_filename.empty() || w->mark_iter >= 0 || w->stgs.verbose) return false;  // (a verbose member prints its own table: solved by scs_solve)
    if (w->n_psd_big > 0 || w->n_cs > 0 || w->n_soc_big > 0 || w->cone.bsize > kBoxMultiMin) return false;
    if (w->aa.mem > 0 && !w->aa.tsqr) return false;
    return true;
  }
  static bool same_shape(const ScsHipWork *a, const ScsHipWork *b) {
    const HostCone &x = a->cone, &y = b->cone;
    return a->device == b->device && a->n == b->n && a->m == b->m && a->has_P == b->has_P && a->normalized == b->normalized &&
           a->linsys == b->linsys &&
           x.z == y.z && x.l == y.l && x.bsize == y.bsize && x.ep == y.ep && x.ed == y.ed && x.q == y.q && x.s == y.s &&
           x.p.size() == y.p.size() && a->aa.mem == b->aa.mem && a->aa.type1 == b->aa.type1 &&
           a->stgs.acceleration_interval == b->stgs.acceleration_interval;
  }

  int vb(long nelem) const { return vec_blocks(nelem); }
  // For the duration of the grouped solve a member's flag block (vec.hpp F_*) is its slice of ONE array, so the host
  // reads every member's CG flags with a single copy and no gather launch
  int *fl_of(int g) const { return flags_d.p + (size_t)g * F_COUNT; }

  // ---- lists: a ring of device slots; a slot is not reused before the host has synchronised at least once
  const int *upload_list(const std::vector<int> &v) {
    if (v.empty()) return nullptr;
    if (++lists_since_sync >= kListSlots) sync();  // (never in practice: every iteration synchronises)
    int *dst = lists_d.p + (size_t)list_slot * G, *stage = lists_h + (size_t)list_slot * G;
    list_slot = (list_slot + 1) % kListSlots;
    std::copy(v.begin(), v.end(), stage);
    HIP_CHECK(hipMemcpyAsync(dst, stage, sizeof(int) * v.size(), hipMemcpyHostToDevice, s));
    return dst;
  }
  // the long-lived list of unfinished members has a buffer of its own (stream order protects the device copy; the
  // pinned stage is rewritten at most once per iteration, and every iteration synchronises)
  void upload_active() {
    std::copy(active.begin(), active.end(), active_h);
    if (!active.empty())
      HIP_CHECK(hipMemcpyAsync(active_buf.p, active_h, sizeof(int) * active.size(), hipMemcpyHostToDevice, s));
    active_d = active_buf.p;
  }
  void sync() {
    HIP_CHECK(hipGetLastError());  // (launches are not checked one by one)
    HIP_CHECK(hipStreamSynchronize(s));
    lists_since_sync = 0;
    iters_since_sync = 0;
    ++syncs;
  }
  template <class T> void go(const T &t, const int *list, int count) {
    t.launch(list, count, s);
    if (t.used && count > 0) ++launches;
  }

  // ---- construction of the argument records
  static CsrView csr_of(const DeviceCsr &M) {
    CsrView V = M.view().csr;
    V.pstride = V.nblk;  // partial slots: this member's own row-block count (the grid is the group's largest)
    return V;
  }
  // records that hold R_x / R_y (they follow `scale`)
  void set_scale_records(int g) {
    ScsHipWork *w = W[(size_t)g];
    const CsrView Ar = csr_of(w->Ar);
    int *fl = fl_of(g);
    t_spmv_y.set(g, Ar, w->ws.p, EpiY{w->ut.p + n, w->rdy(), w->v.p + n}, nullptr, nullptr);
    t_spmv_a.set(g, Ar, w->cg_p.p, EpiDivR{w->tmp_m.p, w->rdy()}, fl + F_DONE, fl + F_STEP);
    if (dense) t_dense_y.set(g, Ar, w->ut.p, EpiY{w->ut.p + n, w->rdy(), w->v.p + n}, nullptr, nullptr);
    t_set_diag_r.set(g, w->diag_r.p, n, m, w->cone.z, w->stgs.rho_x, w->scale);
  }
  void set_aa_update_record(int g, int idx) {
    ScsHipWork *w = W[(size_t)g];
    DeviceAa &a = w->aa;
    t_aa_update.set(g, w->v_prev.p, w->v.p, a.x.p, a.f.p, a.gprev.p, a.S.p, a.Y.p, a.D.p, a.dim, idx, a.npart.p);
  }

  void build() {
    G = (int)W.size();
    ScsHipWork *w0 = W[0];
    n = w0->n; m = w0->m; l = w0->l; has_P = w0->has_P;
    mem = w0->aa.mem; interval = w0->stgs.acceleration_interval;
    dense = w0->dense();
    dn_NP = w0->dn_NP;
    HIP_CHECK(hipHostMalloc((void **)&params_h, sizeof(double) * P_COUNT * G * kParamRing));
    HIP_CHECK(hipHostMalloc((void **)&res_h, sizeof(double) * kResRec * G));
    HIP_CHECK(hipHostMalloc((void **)&aa_res_h, sizeof(double) * AA_R_COUNT * G));
    HIP_CHECK(hipHostMalloc((void **)&flags_h, sizeof(int) * F_COUNT * G));
    HIP_CHECK(hipHostMalloc((void **)&lists_h, sizeof(int) * kListSlots * G));
    HIP_CHECK(hipHostMalloc((void **)&active_h, sizeof(int) * G));
    active_buf.alloc_zero((size_t)G, s);
    params_d.alloc_zero((size_t)P_COUNT * G, s);
    res_d.alloc_zero((size_t)kResRec * G, s);
    aa_res_d.alloc_zero((size_t)AA_R_COUNT * G, s);
    flags_d.alloc_zero((size_t)F_COUNT * G, s);
    lists_d.alloc_zero((size_t)kListSlots * G, s);
    cg_res_min.assign((size_t)G, 0.0);
    aa_mode.assign((size_t)G, 0);
    aa_len.assign((size_t)G, 0);

    const int nbl = vb(l), nbl1 = vb(l - 1), nbn = vb(n), nbm = vb(m), nbnm = vb((long)n + m), nb_admm = vb(std::max(n, m));
    const HostCone &c0 = w0->cone;
    auto each = [&](auto &&fn) { for (int g = 0; g < G; ++g) fn(g, W[(size_t)g]); };
    auto size_all = [&](auto &...t) { (t.resize((size_t)G), ...); };
    size_all(t_sumsq, t_prep, t_spmv_y, t_spmv_r0, t_fin_head, t_spmv_a, t_spmv_at, t_cg_update[0], t_cg_update[1], t_cg_dir[0],
             t_cg_dir[1], t_tau_dots, t_cone_pre, t_v_update, t_rsk, t_res_pri, t_res_dual, t_fin_multi_p, t_fin_multi_d,
             t_gather_res, t_gather_fl, t_set_diag_r, t_precond, t_g_rhs, t_kkt_prep, t_spmv_rhs, t_zero_part, t_fin_tol,
             t_cg_init, t_fin_cg_init, t_zero_iters, t_kkt_y, t_copy_g, t_gg, t_fin_gg, t_v_rescale, t_spmv_ax);
    if (has_P) size_all(t_spmv_pws, t_spmv_p, t_res_px);
    if (dense) size_all(t_dense_rhs, t_dense_gemv, t_dense_gemv_kkt, t_dense_y, t_symv_tiles, t_symv_sum, t_symv_sum_kkt);
    if (c0.bsize > 0) t_box.resize((size_t)G);
    soc_psd_fused = w0->soc_psd_one_launch && w0->n_soc > 0 && w0->n_psd > 0 && !w0->psd_small_one_wave;  // (member_ok: nothing big)
    if (soc_psd_fused) t_soc_psd.resize((size_t)G);
    else {
      if (w0->n_soc > 0) t_soc.resize((size_t)G);
      if (w0->n_psd > 0) t_psd.resize((size_t)G);
    }
    if (c0.ep > 0) t_exp_p.resize((size_t)G);
    if (c0.ed > 0) t_exp_d.resize((size_t)G);
    if (!c0.p.empty()) t_pow.resize((size_t)G);
    std::vector<DeviceAa::TsqrLevel> lv0;
    if (mem > 0) {
      size_all(t_aa_seed, t_aa_update, t_aa_solve, t_aa_apply, t_aa_diffsq, t_fin_safe, t_aa_restore, t_gather_aa);
      lv0 = w0->aa.tsqr_levels(mem);
      n_tsqr = (int)lv0.size();
      if (n_tsqr > kMaxTsqrLevels) throw std::runtime_error("grouped solve: unexpected TSQR depth");
      for (size_t k = 0; k < lv0.size(); ++k) {
        tsqr_fast[k] = lv0[k].fast;
        if (tsqr_fast[k]) {
          const int gxf = (int)((lv0[k].nw + kAaFastWaves - 1) / kAaFastWaves);
          if (tsqr_fast[k] == 1) { t_aa_tsqr_f21[k].resize((size_t)G); t_aa_tsqr_f21[k].gx = gxf; }
          else { t_aa_tsqr_f11[k].resize((size_t)G); t_aa_tsqr_f11[k].gx = gxf; }
          continue;
        }
        t_aa_tsqr[k].resize((size_t)G);
        t_aa_tsqr[k].gx = (int)lv0[k].nw;
        t_aa_tsqr[k].lds = lv0[k].lds;
      }
    }
    int gx_ar = 0, gx_at = 0, gx_pf = 0;
    each([&](int g, ScsHipWork *w) {
      const CsrView Ar = csr_of(w->Ar), At = csr_of(w->At);
      const CsrView Pf = has_P ? csr_of(w->Pf) : CsrView{};
      gx_ar = std::max(gx_ar, Ar.nblk); gx_at = std::max(gx_at, At.nblk); gx_pf = std::max(gx_pf, Pf.nblk);
      double *par = params_d.p + (size_t)g * P_COUNT;
      int *fl = fl_of(g);
      double *uy = w->u.p + n;
      const int *nostall = nullptr;
      t_sumsq.set(g, w->v.p, l, w->part_v.p);
      t_prep.set(g, w->v.p, w->v_prev.p, w->ut.p, w->ws.p, w->u.p, w->g.p, w->diag_r.p, n, m, par, w->part_v.p, nbl, w->sc.p,
                 w->part2.p, nostall);
      set_scale_records(g);
      if (dense) {
        t_dense_rhs.set(g, At, w->v.p + n, EpiDenseRhs{w->cg_b.p, w->rdx(), w->v.p}, nullptr, nullptr);
        t_dense_gemv.set(g, w->dn_G.p, dn_NP, n, w->cg_b.p, w->ut.p, nostall);
        t_dense_gemv_kkt.set(g, w->dn_G.p, dn_NP, n, w->cg_b.p, w->ws.p, nostall);
        t_symv_tiles.set(g, w->dn_G.p, dn_NP, n, w->cg_b.p, w->dn_part.p, nostall);
        t_symv_sum.set(g, w->dn_part.p, dn_NP, n, w->ut.p, nostall);
        t_symv_sum_kkt.set(g, w->dn_part.p, dn_NP, n, w->ws.p, nostall);
      }
      if (has_P) t_spmv_pws.set(g, Pf, w->ws.p, EpiStore{w->cg_Gp.p, 0}, nullptr, nullptr);
      t_spmv_r0.set(g, At, w->ut.p + n,
                    EpiR0{w->cg_r.p, w->cg_p.p, w->cg_M.p, w->rdx(), w->v.p, w->ws.p, has_P ? w->cg_Gp.p : nullptr, w->part.p},
                    nullptr, nullptr);
      t_fin_head.set(g, w->part2.p, nbl, w->part.p, At.nblk, par, w->sc.p, fl, w->ut.p, (long)n + m, nostall);
      if (has_P) t_spmv_p.set(g, Pf, w->cg_p.p, EpiStore{w->cg_Gp.p, 0}, fl + F_DONE, nullptr);
      t_spmv_at.set(g, At, w->tmp_m.p, EpiGp{w->cg_Gp.p, w->cg_p.p, w->rdx(), has_P ? 1 : 0, w->part.p, nullptr}, fl + F_DONE,
                    nullptr);
      t_cg_update[0].set(g, w->ut.p, w->cg_r.p, w->cg_p.p, w->cg_Gp.p, w->cg_M.p, n, w->ut.p + n, w->tmp_m.p, m, w->part.p,
                         At.nblk, w->sc.p, fl, w->part2.p, nullptr);
      t_cg_update[1].set(g, w->ws.p, w->cg_r.p, w->cg_p.p, w->cg_Gp.p, w->cg_M.p, n, nullptr, w->tmp_m.p, m, w->part.p, At.nblk,
                         w->sc.p, fl, w->part2.p, nullptr);
      t_cg_dir[0].set(g, w->cg_p.p, w->cg_r.p, w->cg_M.p, n, w->part2.p, nb_admm, w->sc.p, fl);
      t_cg_dir[1].set(g, w->cg_p.p, w->cg_r.p, w->cg_M.p, n, w->part2.p, nbn, w->sc.p, fl);
      t_tau_dots.set(g, w->ut.p, w->v.p, w->g.p, w->diag_r.p, l - 1, w->part.p, nullptr);
      t_cone_pre.set(g, w->ut.p, w->u.p, w->v.p, w->g.p, n, m, w->cone.z, w->cone.l, par, w->sc.p, w->part.p, nbl1, w->diag_r.p,
                     nullptr);
      if (t_box.used) t_box.set(g, uy + w->cone.off_box, w->box_bl.p, w->box_bu.p, w->cone.bsize, w->sc.p + S_BOX_T, 1, nostall);
      if (t_soc_psd.used)
        t_soc_psd.set(g, uy, w->soc_off.p, w->soc_dim.p, w->n_soc, w->soc_G, soc_wave_blocks(w->n_soc, w->soc_G),
                      PsdBatch{w->psd_off.p, w->psd_order.p, w->psd_woff.p, w->n_psd}, w->psd_scratch.p, w->psd_warm, nostall,
                      (const double *)(par + P_PSD_TOL2));
      if (t_soc.used) t_soc.set(g, uy, w->soc_off.p, w->soc_dim.p, w->n_soc, w->soc_G, nostall);
      if (t_psd.used)
        t_psd.set(g, uy, PsdBatch{w->psd_off.p, w->psd_order.p, w->psd_woff.p, w->n_psd}, w->psd_scratch.p, w->psd_warm, nostall,
                  (const double *)(par + P_PSD_TOL2));
      if (t_exp_p.used) t_exp_p.set(g, uy + w->cone.off_ep, w->cone.ep, 0, nostall);
      if (t_exp_d.used) t_exp_d.set(g, uy + w->cone.off_ed, w->cone.ed, 1, nostall);
      if (t_pow.used) t_pow.set(g, uy + w->cone.off_p, w->pow_a.p, (int)w->cone.p.size(), nostall);
      t_v_update.set(g, w->v.p, w->u.p, w->ut.p, w->stgs.alpha, l, w->part_v.p, nostall);
      t_rsk.set(g, w->rsk.p, w->v.p, w->u.p, w->ut.p, w->diag_r.p, l);
      const double *tau_ptr = w->u.p + (l - 1);
      t_res_pri.set(g, Ar, w->u.p,
                    EpiResPri{w->rsk.p + n, w->h.p + n, w->normalized ? w->Dinv.p : nullptr, tau_ptr, uy, w->part.p}, nullptr,
                    nullptr);
      t_fin_multi_p.set(g, w->part.p, Ar.nblk, 3, 6, w->out.p);
      if (has_P) t_res_px.set(g, Pf, w->u.p, EpiStore{w->px.p, 0}, nullptr, nullptr);
      t_res_dual.set(g, At, uy,
                     EpiResDual{has_P ? w->px.p : nullptr, w->h.p, w->normalized ? w->Einv.p : nullptr, w->u.p, tau_ptr, w->part.p},
                     nullptr, nullptr);
      t_fin_multi_d.set(g, w->part.p, At.nblk, 4, 6, w->out.p + 16);
      t_gather_res.set(g, w->out.p, tau_ptr, w->rsk.p + (l - 1), res_d.p + (size_t)g * kResRec);
      t_gather_fl.set(g, w->fl.p, fl, (int)F_COUNT);  // (once, at the start: the member's own flag block -> the group's)
      // scale update
      t_precond.set(g, w->At.rowptr.p, w->At.col.p, w->At.val.p, w->diag_r.p, has_P ? w->Pdiag.p : nullptr, w->cg_M.p, n);
      t_g_rhs.set(g, w->g.p, w->h.p, n, m);
      t_kkt_prep.set(g, w->g.p, w->diag_r.p, w->tmp_m.p, n, m);
      t_spmv_rhs.set(g, At, w->tmp_m.p, EpiRhs{w->cg_b.p, w->g.p}, nullptr, nullptr);
      t_zero_part.set(g, w->part.p, 0.0, 1);
      t_fin_tol.set(g, w->part.p, 1, 0.0, 1.0, 1e-12, 0, nullptr, w->sc.p, fl);
      t_cg_init.set(g, w->cg_b.p, w->cg_Gp.p, nullptr, w->cg_M.p, w->ws.p, w->cg_r.p, w->cg_p.p, n, 0, fl, w->part.p, nullptr);
      t_fin_cg_init.set(g, w->part.p, nbn, 0, w->sc.p, fl);
      t_zero_iters.set(g, fl + F_ITERS, 0, 1);
      t_spmv_ax.set(g, Ar, w->ws.p, EpiStore{w->tmp_m.p, 0}, nullptr, nullptr);
      t_kkt_y.set(g, w->g.p, w->tmp_m.p, w->diag_r.p, n, m);
      t_copy_g.set(g, w->ws.p, w->g.p, (long)n);
      t_gg.set(g, w->g.p, w->diag_r.p, l - 1, w->part.p);
      t_fin_gg.set(g, w->part.p, nbl1, w->sc.p, (int)S_GG);
      t_v_rescale.set(g, w->v.p, w->rsk.p, w->u.p, w->ut.p, w->diag_r.p, l);
      if (mem > 0) {
        DeviceAa &a = w->aa;
        const int nb = a.nbl();
        t_aa_seed.set(g, w->v_prev.p, w->v.p, a.x.p, a.f.p, a.gprev.p, a.dim);
        set_aa_update_record(g, 0);
        const std::vector<DeviceAa::TsqrLevel> lv = a.tsqr_levels(mem);
        for (size_t k = 0; k < lv.size(); ++k) {
          if (tsqr_fast[k] == 1) t_aa_tsqr_f21[k].set(g, lv[k].W, lv[k].tiles_per_wave, lv[k].nw, lv[k].out, lv[k].out_ld);
          else if (tsqr_fast[k] == 2) t_aa_tsqr_f11[k].set(g, lv[k].W, lv[k].tiles_per_wave, lv[k].nw, lv[k].out, lv[k].out_ld);
          else t_aa_tsqr[k].set(g, lv[k].W, lv[k].rho, lv[k].tiles_per_wave, lv[k].out, lv[k].out_ld);
        }
        t_aa_solve.set(g, lv.back().out, mem, a.ncols(), a.type1, a.regularization, a.max_weight_norm, a.npart.p, nb, a.res.p);
        t_aa_apply.set(g, w->v.p, a.D.p, a.S.p, a.x.p, a.res.p + AA_R_GAMMA, a.dim, mem, a.relaxation, a.res.p + AA_R_OK);
        t_gather_aa.set(g, a.res.p, aa_res_d.p + (size_t)g * AA_R_COUNT, (long)AA_R_COUNT);
        t_aa_diffsq.set(g, w->v_prev.p, w->v.p, a.dim, a.spart.p);
        t_fin_safe.set(g, a.spart.p, nb, a.safeguard_factor, a.res.p, fl + F_SAFE_BAD);
        t_aa_restore.set(g, w->v.p, w->v_prev.p, a.f.p, a.x.p, a.dim, fl + F_SAFE_BAD);
      }
    });
    // launch geometry: exactly what the one-problem launches use
    t_sumsq.gx = t_prep.gx = t_cone_pre.gx = t_v_update.gx = t_rsk.gx = t_set_diag_r.gx = t_v_rescale.gx = nbl;
    t_spmv_y.gx = t_spmv_a.gx = t_res_pri.gx = t_spmv_ax.gx = t_dense_y.gx = gx_ar;
    t_spmv_r0.gx = t_spmv_at.gx = t_res_dual.gx = t_spmv_rhs.gx = t_dense_rhs.gx = gx_at;
    t_dense_gemv.gx = t_dense_gemv_kkt.gx = dense_gemv_blocks(n);
    t_symv_tiles.gx = dense ? dense_symv_tiles(dn_NP) : 1;
    t_symv_sum.gx = t_symv_sum_kkt.gx = ceil_div(n, kDenseThreads);
    if (dense) {  // the members' matrices as the batched factorisation kernels take them
      std::vector<DenseMat> hm((size_t)G);
      std::vector<DenseSrc> hs((size_t)G);
      for (int g = 0; g < G; ++g) { hm[(size_t)g] = W[(size_t)g]->dense_mat(); hs[(size_t)g] = W[(size_t)g]->dense_src(); }
      dmat_d.upload(hm.data(), (size_t)G, s);
      dsrc_d.upload(hs.data(), (size_t)G, s);
      HIP_CHECK(hipStreamSynchronize(s));  // (hm / hs are locals)
    }
    t_spmv_pws.gx = t_spmv_p.gx = t_res_px.gx = gx_pf;
    t_cg_update[0].gx = nb_admm;
    t_cg_update[1].gx = t_cg_dir[0].gx = t_cg_dir[1].gx = t_cg_init.gx = t_precond.gx = t_copy_g.gx = nbn;
    t_tau_dots.gx = t_gg.gx = nbl1;
    t_g_rhs.gx = nbnm;
    t_kkt_prep.gx = t_kkt_y.gx = nbm;
    t_soc.gx = soc_wave_blocks(w0->n_soc, w0->soc_G);
    t_psd.gx = w0->n_psd;
    t_soc_psd.gx = t_soc.gx + t_psd.gx;
    t_exp_p.gx = ceil_div(c0.ep, kConeThreads);
    t_exp_d.gx = ceil_div(c0.ed, kConeThreads);
    t_pow.gx = ceil_div((long)c0.p.size(), kConeThreads);
    if (mem > 0) t_aa_seed.gx = t_aa_update.gx = t_aa_apply.gx = t_aa_diffsq.gx = t_aa_restore.gx = w0->aa.nbl();
    upload_all();
    {  // seed the group's flag array from the members' own blocks (the step parity F_STEP carries over)
      std::vector<int> all((size_t)G);
      std::iota(all.begin(), all.end(), 0);
      go(t_gather_fl, upload_list(all), G);
    }
    sync();  // (records may be rewritten from here on)
  }

  template <class F> void for_tables(F &&f) {
    f(t_sumsq); f(t_prep); f(t_spmv_y); f(t_spmv_pws); f(t_spmv_p); f(t_res_px); f(t_spmv_ax); f(t_spmv_r0); f(t_fin_head);
    f(t_spmv_a); f(t_spmv_at); f(t_cg_update[0]); f(t_cg_update[1]); f(t_cg_dir[0]); f(t_cg_dir[1]); f(t_tau_dots);
    f(t_cone_pre); f(t_box); f(t_soc); f(t_psd); f(t_soc_psd); f(t_exp_p); f(t_exp_d); f(t_pow); f(t_v_update); f(t_rsk); f(t_res_pri);
    f(t_res_dual); f(t_fin_multi_p); f(t_fin_multi_d); f(t_gather_res); f(t_gather_fl); f(t_set_diag_r); f(t_precond);
    f(t_g_rhs); f(t_kkt_prep); f(t_spmv_rhs); f(t_zero_part); f(t_fin_tol); f(t_cg_init); f(t_fin_cg_init); f(t_zero_iters);
    f(t_kkt_y); f(t_copy_g); f(t_gather_aa); f(t_gg); f(t_fin_gg); f(t_v_rescale); f(t_aa_seed); f(t_aa_update);
    for (int k = 0; k < n_tsqr; ++k) { f(t_aa_tsqr_f21[k]); f(t_aa_tsqr_f11[k]); f(t_aa_tsqr[k]); }
    f(t_aa_solve); f(t_aa_apply); f(t_aa_diffsq); f(t_fin_safe); f(t_aa_restore);
    f(t_dense_rhs); f(t_dense_gemv); f(t_dense_gemv_kkt); f(t_dense_y); f(t_symv_tiles); f(t_symv_sum); f(t_symv_sum_kkt);
  }
  void upload_all() {
    for_tables([&](auto &t) { t.upload(s); });
  }

  // ---- pieces of the iteration, each over a sub-list of the group ----
  void read_flags(const int *, int) {  // enqueue: every member's flag block -> flags_h
    HIP_CHECK(hipMemcpyAsync(flags_h, flags_d.p, sizeof(int) * F_COUNT * G, hipMemcpyDeviceToHost, s));
  }
  const int *flags_of(int g) const { return flags_h + (size_t)g * F_COUNT; }
  void process_pending_flags(int g) {  // ScsHipWork::process_pending_flags on the group's copy
    ScsHipWork *w = W[(size_t)g];
    if (w->aa.pending_safeguard) {
      const bool bad = flags_of(g)[F_SAFE_BAD] != 0;
      w->aa.safeguard_verdict(bad);
      if (bad) w->rejected_accel++;
      else w->accepted_accel++;
    }
  }
  void cg_step(const int *list, int count, int variant) {
    go(t_spmv_a, list, count);
    if (has_P) go(t_spmv_p, list, count);
    go(t_spmv_at, list, count);
    go(t_cg_update[variant], list, count);
    go(t_cg_dir[variant], list, count);
  }
  // CG steps to enqueue for a set of members whose predicted remaining step counts are `need`: enough for three
  // quarters of them.  A step enqueued for a member that is already done is a launch of early-exit workgroups — ~20 us
  // for 512 members — while a second round for the slow quarter costs one host synchronisation (~40 us) and runs on a
  // quarter of the grid.  (A single member gets exactly its prediction.)
  static int chunk_for(std::vector<int> &need) {
    if (need.empty()) return 2;
    std::sort(need.begin(), need.end());
    const int c = need[std::min(need.size() - 1, (need.size() * 3) / 4)];
    return std::max(2, std::min(c, 64));
  }
  // run the CG of the listed members to the end (their start is already enqueued); returns with flags_h holding every
  // listed member's final flags.  predict(g) = steps member g is expected to need in total.
  void finish_cg(const std::vector<int> &members, const int *list_d, int variant, const std::function<int(int)> &predict,
                 const std::function<void()> &after_first_sync) {
    std::vector<int> need;
    for (int g : members) need.push_back(predict(g));
    const int first_chunk = std::max(1, std::min(chunk_for(need), 10 * n));
    for (int k = 0; k < first_chunk; ++k) cg_step(list_d, (int)members.size(), variant);
    read_flags(list_d, (int)members.size());
    sync();
    if (after_first_sync) after_first_sync();
    auto collect = [&](const std::vector<int> &from) {
      std::vector<int> out;
      for (int g : from)
        if (!flags_of(g)[F_DONE] && flags_of(g)[F_ITERS] < 10 * n) out.push_back(g);
      return out;
    };
    std::vector<int> nd = collect(members);
    while (!nd.empty()) {
      need.clear();
      for (int g : nd) {
        const int done = flags_of(g)[F_ITERS];
        need.push_back(std::max(predict(g) - done, std::max(done / 2, 2)));  // (a wrong prediction: grow geometrically)
      }
      //  the device steps are only gated by F_DONE: never enqueue a member past the 10 n cap run_cg stops at exactly
      int most_done = 0;
      for (int g : nd) most_done = std::max(most_done, flags_of(g)[F_ITERS]);
      const int chunk = std::max(1, std::min(chunk_for(need), 10 * n - most_done));
      const int *ld = upload_list(nd);
      for (int k = 0; k < chunk; ++k) cg_step(ld, (int)nd.size(), variant);
      read_flags(ld, (int)nd.size());
      sync();
      nd = collect(nd);
    }
  }

  void enqueue_cones(const int *list, int count) {
    go(t_box, list, count);
    go(t_soc_psd, list, count);
    go(t_soc, list, count);
    go(t_psd, list, count);
    go(t_exp_p, list, count);
    go(t_exp_d, list, count);
    go(t_pow, list, count);
  }

  // adaptive-scale update of the listed members (ScsHipWork::update_scale after a positive decision); first_setup: the deferred end
  // of scs_init of dense members (ScsHipWork::finish_pending_setup) — the same launches without the iterate rescaling
  void apply_scale_updates(const std::vector<int> &su, bool first_setup = false) {
    for (int g : su) set_scale_records(g);
    t_spmv_y.upload(s); t_spmv_a.upload(s); t_set_diag_r.upload(s); t_dense_y.upload(s);
    const int *ld = upload_list(su);
    const int cnt = (int)su.size();
    go(t_set_diag_r, ld, cnt);
    if (dense) {  // ScsHipWork::set_diag_r + update_work_cache with the dense direct solve: re-form and re-invert G, g = KKT^-1 [c; -b]
      dense_factor_group(dsrc_d.p, dmat_d.p, ld, cnt, dn_NP, s);
      launches += 1 + 3 * (dn_NP / kDenseB);
      for (int g : su) W[(size_t)g]->dense_factorisations++;
      go(t_g_rhs, ld, cnt);
      go(t_kkt_prep, ld, cnt);
      go(t_spmv_rhs, ld, cnt);
      go_dense_gemv(ld, cnt, true);
      go(t_spmv_ax, ld, cnt);
      go(t_kkt_y, ld, cnt);
      go(t_copy_g, ld, cnt);
      go(t_gg, ld, cnt);
      go(t_fin_gg, ld, cnt);
      if (first_setup) return;
      for (int g : su) W[(size_t)g]->aa.reset();
      go(t_v_rescale, ld, cnt);
      for (int g : su) W[(size_t)g]->v_norm_fresh = false;
      return;
    }
    go(t_precond, ld, cnt);
    // update_work_cache: g = (R + M)^{-1} [c; -b] by a cold PCG to 1e-12, then g'Rg
    go(t_g_rhs, ld, cnt);
    go(t_kkt_prep, ld, cnt);
    go(t_spmv_rhs, ld, cnt);
    go(t_zero_part, ld, cnt);
    go(t_fin_tol, ld, cnt);
    go(t_cg_init, ld, cnt);
    go(t_fin_cg_init, ld, cnt);
    go(t_zero_iters, ld, cnt);
    finish_cg(su, ld, 1, [&](int g) { return std::min(4 * W[(size_t)g]->last_cg_iters + 8, 64); }, nullptr);  // (cold, to 1e-12)
    for (int g : su) {
      ScsHipWork *w = W[(size_t)g];
      w->last_cg_iters = flags_of(g)[F_ITERS];
      w->tot_cg_iters += w->last_cg_iters;
    }
    go(t_spmv_ax, ld, cnt);
    go(t_kkt_y, ld, cnt);
    go(t_copy_g, ld, cnt);
    go(t_gg, ld, cnt);
    go(t_fin_gg, ld, cnt);
    if (first_setup) return;
    for (int g : su) W[(size_t)g]->aa.reset();
    go(t_v_rescale, ld, cnt);
    for (int g : su) W[(size_t)g]->v_norm_fresh = false;
  }

  // ---- the lock-step loop ----
  void run(int warm_start) {
    {  // members fresh from scs_init: R, G^{-1} (dense) or the preconditioner (indirect) and g for all of them in one grouped pass
      std::vector<int> pend;
      for (int g = 0; g < G; ++g) {
        if (W[(size_t)g]->setup_failed) throw std::runtime_error(W[(size_t)g]->setup_failed_msg(g));
        if (W[(size_t)g]->setup_pending) pend.push_back(g);
      }
      if (!pend.empty()) {
        const double t0 = now_ms();
        apply_scale_updates(pend, /*first_setup=*/true);
        std::vector<double> gg(pend.size(), 0.);
        if (dense)  //  the unpivoted Gauss-Jordan sweep checks nothing on the way: g' R g of every member must be finite,
                    // exactly as ScsHipWork::finish_pending_setup asks of a member that finishes its setup alone
          for (size_t k = 0; k < pend.size(); ++k)
            HIP_CHECK(hipMemcpyAsync(&gg[k], W[(size_t)pend[k]]->sc.p + S_GG, sizeof(double), hipMemcpyDeviceToHost, s));
        sync();
        const double each = (now_ms() - t0) / (double)pend.size();
        int bad = -1;
        for (size_t k = 0; k < pend.size(); ++k) {
          ScsHipWork *w = W[(size_t)pend[k]];
          w->setup_time += each;  // (the deferred part of scs_init: counted as setup, not as solve time)
          if (dense && !std::isfinite(gg[k])) {
            w->setup_failed = true;  // stays pending + failed: any later solve of this workspace reports the same error
            if (bad < 0) bad = pend[k];
          } else {
            w->setup_pending = false;
          }
        }
        if (bad >= 0) throw std::runtime_error(W[(size_t)bad]->setup_failed_msg(bad));
      }
    }
    t_start = now_ms();  // behind the deferred setup: solve_time is the solve (ScsInfo as the reference fills it)
    mr_allowed_saved.assign((size_t)G, 0);
    for (int g = 0; g < G; ++g) {
#ifdef SCS_HIP_LABS
      mr_allowed_saved[(size_t)g] = W[(size_t)g]->mr_allowed ? 1 : 0;
      W[(size_t)g]->mr_allowed = false;  // the grouped loop drives PCG steps through its own tables (minres.hpp: one workspace at a time)
      W[(size_t)g]->mr_active = false;   // (a member that had switched to MINRES in a solve of its own runs PCG here, and may switch again later)
#endif
      W[(size_t)g]->begin_solve(sols[(size_t)g], infos[(size_t)g], warm_start);
    }
    for (int g = 0; g < G; ++g) {
      if (dense)
        std::snprintf(infos[(size_t)g]->lin_sys_solver, sizeof(infos[(size_t)g]->lin_sys_solver),
                      "dense-direct HIP gfx950 (explicit inverse of the reduced KKT matrix, order %d; fp64 MFMA Gauss-Jordan; grouped solve of %d)", n, G);
      else
        std::snprintf(infos[(size_t)g]->lin_sys_solver, sizeof(infos[(size_t)g]->lin_sys_solver),
                      "sparse-indirect HIP gfx950 (CSR-stream SpMV, PCG; grouped solve of %d)", G);
      set_scale_records(g);  // (`scale` starts from the settings again)
    }
    t_spmv_y.upload(s); t_spmv_a.upload(s); t_set_diag_r.upload(s); t_dense_y.upload(s);
    active.resize((size_t)G);
    std::iota(active.begin(), active.end(), 0);
    upload_active();
    std::vector<int> tmp_list;
    const bool stats = (opts().debug & DBG_GROUP) != 0;  // SCS_HIP_DEBUG=group
    InterruptListener ctrlc;  // (loop.hpp: Ctrl-C ends every member that is still running with SCS_SIGINT)
    for (int i = 0; !active.empty(); ++i) {
      if (InterruptListener::interrupted()) {
        sync();
        for (int g : active) {
          infos[(size_t)g]->status_val = SCS_SIGINT;
          W[(size_t)g]->finish_solve(sols[(size_t)g], infos[(size_t)g], i, t_start, t_lin, t_cone, t_acc, /*grouped=*/true);
        }
        active.clear();
        break;
      }
      ++lockstep_iters;
      const int na = (int)active.size();
      if (stats && (i % 200 == 0 || (i < 200 && i % 50 == 0)))
        std::fprintf(stderr, "[scs-hip group] iteration %5d: %4d active, %8.1f ms, %ld launches, %d syncs\n", i, na, now_ms() - t_start, launches, syncs);
      const bool aa_now = mem > 0 && i > 0 && (i % interval == 0);
      double t = now_ms();
      std::vector<int> aa_solved;
      if (aa_now) {
        bool pending = false;
        for (int g : active) pending = pending || W[(size_t)g]->aa.pending_safeguard;
        if (pending) {  // acceleration_interval == 1: last step's safeguard verdict decides this step's history
          read_flags(active_d, na);
          sync();
          for (int g : active) process_pending_flags(g);
        }
        std::vector<int> seed, upd;
        for (int g : active) {
          int len = 0, idx = 0;
          const int mode = W[(size_t)g]->aa.plan(len, idx);
          aa_mode[(size_t)g] = mode; aa_len[(size_t)g] = len;
          W[(size_t)g]->aa_norm = 0;
          if (mode == 1) seed.push_back(g);
          if (mode >= 2) { upd.push_back(g); set_aa_update_record(g, idx); }
          if (mode == 3) aa_solved.push_back(g);
        }
        if (!seed.empty()) go(t_aa_seed, upload_list(seed), (int)seed.size());
        if (!upd.empty()) {
          t_aa_update.upload(s);
          go(t_aa_update, upload_list(upd), (int)upd.size());
        }
        if (!aa_solved.empty()) {
          const int *ld = upload_list(aa_solved);
          const int cnt = (int)aa_solved.size();
          for (int k = 0; k < n_tsqr; ++k) { go(t_aa_tsqr_f21[k], ld, cnt); go(t_aa_tsqr_f11[k], ld, cnt); go(t_aa_tsqr[k], ld, cnt); }
          go(t_aa_solve, ld, cnt);
          go(t_aa_apply, ld, cnt);
          go(t_gather_aa, ld, cnt);
          HIP_CHECK(hipMemcpyAsync(aa_res_h, aa_res_d.p, sizeof(double) * AA_R_COUNT * G, hipMemcpyDeviceToHost, s));
          // whether a step was taken is only known after the read-back below: the norm of v is recomputed either
          // way (k_sumsq of an unchanged v leaves the bits k_v_update left)
          for (int g : aa_solved) W[(size_t)g]->v_norm_fresh = false;
        }
        t_acc += now_ms() - t;
      }
      // ---- project_lin_sys
      t = now_ms();
      // (the staging slot is rewritten kParamRing iterations later: the dense loop synchronises at least that often)
      double *const params_slot = params_h + (size_t)(i % kParamRing) * P_COUNT * G;
      if (++iters_since_sync >= kParamRing - 2) { sync(); iters_since_sync = 0; }
      for (int g : active) {
        double *p = params_slot + (size_t)g * P_COUNT;
        p[P_DO_SCALE] = i >= 1 ? 1.0 : 0.0;
        p[P_RES_MIN] = cg_res_min[(size_t)g];
        p[P_IPOW] = std::pow((double)i + 1, 1.5);
        p[P_FIRST] = i < 1 ? 1.0 : 0.0;
        p[P_PSD_TOL2] = W[(size_t)g]->psd_tol2_for(i);
      }
      // (dense linsys: what its kernels read of the block — P_DO_SCALE, P_FIRST, P_PSD_TOL2 — only moves at i <= 1 and at checks /
      //  scale updates / membership changes; a lone straggler's iteration is ~10 launches, the copy would be one more host call)
      bool upload_params = !dense || i <= 2 || params_dirty;
      if (!upload_params)
        for (int g : active) {
          const double *p = params_slot + (size_t)g * P_COUNT, *q = params_last.data() + (size_t)g * P_COUNT;
          if (p[P_DO_SCALE] != q[P_DO_SCALE] || p[P_FIRST] != q[P_FIRST] || p[P_PSD_TOL2] != q[P_PSD_TOL2]) { upload_params = true; break; }
        }
      if (upload_params) {
        HIP_CHECK(hipMemcpyAsync(params_d.p, params_slot, sizeof(double) * P_COUNT * G, hipMemcpyHostToDevice, s));
        params_last.assign(params_slot, params_slot + (size_t)P_COUNT * G);
        params_dirty = false;
      }
      tmp_list.clear();
      for (int g : active)
        if (!W[(size_t)g]->v_norm_fresh) { tmp_list.push_back(g); W[(size_t)g]->v_norm_fresh = true; }
      if (!tmp_list.empty()) go(t_sumsq, tmp_list.size() == active.size() ? active_d : upload_list(tmp_list), (int)tmp_list.size());
      go(t_prep, active_d, na);
      auto deferred_host_work = [&] {
        // first synchronisation of the iteration: everything the host deferred
        for (int g : aa_solved) {
          ScsHipWork *w = W[(size_t)g];
          w->aa_norm = w->aa.complete(aa_res_h + (size_t)g * AA_R_COUNT, aa_len[(size_t)g]);
        }
        if (aa_now)
          for (int g : active)
            if (aa_mode[(size_t)g] >= 1) W[(size_t)g]->aa.iter++;
        for (int g : active) process_pending_flags(g);
      };
      if (dense) {
        // three dependent launches, nothing to wait for: the host only synchronises where it has something to decide
        // (an Anderson step's outcome, a convergence check) — plain iterations are enqueued back to back
        go(t_dense_rhs, active_d, na);
        go_dense_gemv(active_d, na, false);
        go(t_dense_y, active_d, na);
        if (aa_now) {
          read_flags(active_d, na);
          sync();
          iters_since_sync = 0;
          deferred_host_work();
        }
        for (int g : active) W[(size_t)g]->last_cg_iters = 0;
      } else {
      go(t_spmv_y, active_d, na);
      if (has_P) go(t_spmv_pws, active_d, na);
      go(t_spmv_r0, active_d, na);
      go(t_fin_head, active_d, na);
      const int pred_mode = opts().group_predict;  // (labs) 0: max + 1
      finish_cg(active, active_d, 0, [&](int g) { return pred_mode ? W[(size_t)g]->recent_cg_q3() : W[(size_t)g]->recent_cg_max() + 1; }, deferred_host_work);
      iters_since_sync = 0;
      for (int g : active) {
        ScsHipWork *w = W[(size_t)g];
        w->last_cg_iters = flags_of(g)[F_ITERS];
        w->note_cg_iters(w->last_cg_iters);
        w->tot_cg_iters += w->last_cg_iters;
      }
      }
      t_lin += now_ms() - t;
      // ---- tau, cone projections
      t = now_ms();
      go(t_tau_dots, active_d, na);
      go(t_cone_pre, active_d, na);
      enqueue_cones(active_d, na);
      t_cone += now_ms() - t;
      // ---- residuals, termination, adaptive scale
      const bool check = (i % 25 == 0);
      std::vector<int> C, finished, su;
      std::vector<char> stop((size_t)G, 0);  // 1: leaves the loop at this iteration's check (converged / time limit)
      for (int g : active)
        if (check || i == W[(size_t)g]->stgs.max_iters - 1) C.push_back(g);
      if (!C.empty()) {
        const int *ld = C.size() == active.size() ? active_d : upload_list(C);
        const int cnt = (int)C.size();
        go(t_rsk, ld, cnt);
        go(t_res_pri, ld, cnt);
        go(t_fin_multi_p, ld, cnt);
        if (has_P) go(t_res_px, ld, cnt);
        go(t_res_dual, ld, cnt);
        go(t_fin_multi_d, ld, cnt);
        go(t_gather_res, ld, cnt);
        HIP_CHECK(hipMemcpyAsync(res_h, res_d.p, sizeof(double) * kResRec * G, hipMemcpyDeviceToHost, s));
        sync();
        for (int g : C) {
          ScsHipWork *w = W[(size_t)g];
          w->r.last_iter = i;
          w->consume_residuals(res_h + (size_t)g * kResRec);
          if (!check) continue;
          w->note_check_residuals();
          cg_res_min[(size_t)g] = w->cg_res_min;
          if ((infos[(size_t)g]->status_val = w->has_converged(i)) != 0) { stop[(size_t)g] = 1; continue; }
          if (w->stgs.time_limit_secs > 0 && (now_ms() - t_start) > 1e3 * w->stgs.time_limit_secs) { stop[(size_t)g] = 1; continue; }
          if (w->stgs.adaptive_scale && w->decide_scale_update(i)) su.push_back(g);
        }
        if (!su.empty()) apply_scale_updates(su);
      }
      // ---- dual update (not for the members that stop here: solve_impl breaks before it)
      std::vector<int> cont;
      for (int g : active)
        if (!stop[(size_t)g]) cont.push_back(g);
      const int *cont_d = cont.size() == active.size() ? active_d : upload_list(cont);
      go(t_v_update, cont_d, (int)cont.size());
      for (int g : cont) W[(size_t)g]->v_norm_fresh = true;
      if (aa_now) {
        t = now_ms();
        std::vector<int> sg;
        for (int g : cont) {
          ScsHipWork *w = W[(size_t)g];
          if (!w->aa.success) { w->accepted_accel++; continue; }
          w->aa.success = 0;
          w->aa.pending_safeguard = true;
          w->v_norm_fresh = false;
          sg.push_back(g);
        }
        if (!sg.empty()) {
          const int *ld = upload_list(sg);
          go(t_aa_diffsq, ld, (int)sg.size());
          go(t_fin_safe, ld, (int)sg.size());
          go(t_aa_restore, ld, (int)sg.size());
        }
        t_acc += now_ms() - t;
      }
      // ---- members that end here
      std::vector<int> last;
      for (int g : cont)
        if (i == W[(size_t)g]->stgs.max_iters - 1) last.push_back(g);
      {  // solve_impl's finalisation reads the flags once more: a safeguard enqueued just now (`last`) or — dense linsys, where plain
         // iterations do not synchronise — at an Anderson step since the last synchronisation (a member that converges at this check)
        std::vector<int> ending(last);
        for (int g : active)
          if (stop[(size_t)g]) ending.push_back(g);
        bool pending = false;
        for (int g : ending) pending = pending || W[(size_t)g]->aa.pending_safeguard;
        if (pending) {
          read_flags(nullptr, 0);
          sync();
          for (int g : ending) process_pending_flags(g);
        }
      }
      for (int g : active) {
        const bool ends = stop[(size_t)g] || i == W[(size_t)g]->stgs.max_iters - 1;
        if (!ends) continue;
        const int iters = stop[(size_t)g] ? i : i + 1;
        const double tf0 = now_ms();
        W[(size_t)g]->finish_solve(sols[(size_t)g], infos[(size_t)g], iters, t_start, t_lin, t_cone, t_acc, /*grouped=*/true);
        t_finish += now_ms() - tf0;
        finished.push_back(g);
      }
      if (!finished.empty()) {
        std::vector<int> keep;
        for (int g : active)
          if (std::find(finished.begin(), finished.end(), g) == finished.end()) keep.push_back(g);
        active.swap(keep);
        lists_since_sync = 0;  // finish_solve synchronised
        upload_active();
      }
    }
    HIP_CHECK(hipStreamSynchronize(s));
#ifdef SCS_HIP_LABS
    for (int g = 0; g < G; ++g) {  //  what the group took from its members' Krylov state goes back
      ScsHipWork *w = W[(size_t)g];
      w->mr_allowed = mr_allowed_saved[(size_t)g] != 0;
      if (w->mr_ready) w->mr_precond_stale = true;  // scale updates inside the group went through t_set_diag_r / t_precond only
    }
#endif
    if (opts().debug & DBG_GROUP)
      std::fprintf(stderr, "[scs-hip group] members %d, lock-step iterations %d, grouped launches %ld (%.1f per iteration), host syncs %d, %.1f ms (%.1f ms of it finishing members: un-scaling, s'y, downloads)\n",
                   G, lockstep_iters, launches, (double)launches / std::max(lockstep_iters, 1), syncs, now_ms() - t_start, t_finish);
  }
};

}  // namespace scship
