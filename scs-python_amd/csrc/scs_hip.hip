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

namespace scship {

static thread_local std::string g_last_error;
inline void set_last_error(const std::string &s) { g_last_error = s; }
// Device of the NEXT scs_init / standalone entry point: the process default (scs_hip_set_device) unless the calling
// thread has its own (scs_hip_set_thread_device).  A workspace remembers the device it was created on and every later
// call on it selects that device, so one process may drive several GPUs.
static std::atomic<int> g_default_device{0};
static thread_local int t_device = -1;
static int current_device() { return t_device >= 0 ? t_device : g_default_device.load(); }

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------- runtime-object pools
// Measured on this runtime (tools/api_cost.hip): hipStreamCreate 2.55 ms, hipStreamDestroy 1.6 ms, hipHostFree 0.13 ms,
// against 0.7 ms of kernels in the scs_init of a config-5 problem — a batch of 512 small problems spent more time
// creating and destroying streams than solving.  So streams and pinned blocks are pooled per process:
//  * a workspace takes the LEAST-USED stream of its device's pool; the pool grows (up to SCS_HIP_STREAMS, default 32)
//    while every stream has a user, so up to that many live workspaces own a stream each — independent instances run
//    concurrently as before (R:test/test_thread_safety.py:78-93; the device has a handful of hardware queues) — and
//    beyond it streams are shared (stream order keeps every instance correct; the grouped solve puts its members on
//    one stream anyway).  Streams are never destroyed.
//  * one pinned, device-mapped block per workspace holds all its host-side scalars / flags; finished workspaces
//    return their block to a free list.
// Runtime configuration set when this library is loaded (before the HIP runtime reads its flags at the first API call; an
// existing value is kept, SCS_HIP_RUNTIME_ENV=0 leaves the environment alone): GPU_PINNED_MIN_XFER_SIZE (MiB).  Below it
// the runtime stages copies from / to pageable memory through its own pinned buffers; above it it pins the CALLER's pages
// (a userptr registration with the kernel driver), and some time after such pages are released or unmapped the driver
// evicts every queue of this process for 30-80 ms.  Measured (tools/dbg/config2_inflow.py, profiles/r03_queue_eviction.txt):
// a config-2 solve of 40 ms takes 115 ms in ~40 % of the runs that follow another workload's release; ONE hole of 30-80 ms
// between two already-queued kernels in the rocprofv3 trace; ~10 % with the threshold raised (own staging of every transfer or
// hipHostRegister / hipHostUnregister around the copy: 8-28 % / 46 %).  Cost: scs_init of the metric workload 81 -> 88 ms; x, y, s leave through
// a pinned mirror of the workspace instead (download_solution), which is as fast as the pinning path was.
// (priority 101: before this library's own HIP module constructor talks to the runtime; scs/_scs_hip.py and bench.py set the
// same default before they load the runtime at all)
__attribute__((constructor(101))) static void scs_hip_runtime_env() {
  const char *off = getenv("SCS_HIP_RUNTIME_ENV");
  if (!(off && off[0] == '0')) setenv("GPU_PINNED_MIN_XFER_SIZE", "1000000", 0);
}

struct StreamPool {
  struct Dev { std::vector<hipStream_t> streams; std::vector<int> users; };
  std::mutex mtx;
  std::vector<Dev> devs;
  static int cap() {
    static const int c = [] { const char *e = getenv("SCS_HIP_STREAMS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 32; }();  // (process-wide: read once)
    return c;
  }
  hipStream_t acquire(int device, bool *shared) {
    std::lock_guard<std::mutex> g(mtx);
    if ((int)devs.size() <= device) devs.resize((size_t)device + 1);
    Dev &d = devs[(size_t)device];
    int best = -1;
    for (size_t i = 0; i < d.streams.size(); ++i)
      if (best < 0 || d.users[i] < d.users[(size_t)best]) best = (int)i;
    if ((best < 0 || d.users[(size_t)best] > 0) && (int)d.streams.size() < cap()) {
      hipStream_t st = nullptr;
      HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      d.streams.push_back(st);
      d.users.push_back(0);
      best = (int)d.streams.size() - 1;
    }
    *shared = d.users[(size_t)best] > 0;
    d.users[(size_t)best]++;
    return d.streams[(size_t)best];
  }
  void release(int device, hipStream_t st) {
    std::lock_guard<std::mutex> g(mtx);
    if ((int)devs.size() <= device) return;
    Dev &d = devs[(size_t)device];
    for (size_t i = 0; i < d.streams.size(); ++i)
      if (d.streams[i] == st && d.users[i] > 0) { d.users[i]--; return; }
  }
};
static StreamPool g_streams;

constexpr size_t kPinnedBlockBytes = 8192;
struct PinnedPool {
  std::mutex mtx;
  std::vector<void *> free_blocks;
  void *acquire() {
    {
      std::lock_guard<std::mutex> g(mtx);
      if (!free_blocks.empty()) { void *p = free_blocks.back(); free_blocks.pop_back(); return p; }
    }
    void *p = nullptr;
    HIP_CHECK(hipHostMalloc(&p, kPinnedBlockBytes, hipHostMallocMapped));
    return p;
  }
  void release(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mtx);
    if (free_blocks.size() < 4096) { free_blocks.push_back(p); return; }
    (void)hipHostFree(p);
  }
};
static PinnedPool g_pinned;

// ---------------------------------------------------------------- device CSR
struct DeviceCsr {
  DevBuf<int> rowptr, col;
  DevBuf<int4> rowblk;
  DevBuf<double> val;
  int rows = 0, cols = 0, nblk = 0;
  long nnz = 0;
  // optional L2-blocked copy (spmv.hpp) used by the mat-vec kernels when the gather vector exceeds L2
  bool has_slab = false;
  DevBuf<int> s_segptr, s_col, s_perm;  // s_perm: source index of every slab entry in CSR order (-1 = padding)
  DevBuf<unsigned short> s_roff;
  DevBuf<double> s_val;
  int s_nchunks = 0, s_S = 0, s_R = 0, s_max_seg = 0;
  // optional column-sorted pass copy (spmv_cs.hpp); preferred over the slab copy when both could be built
  DeviceCs cs;
  static bool cs_enabled() { return opts().cs; }  // SCS_HIP_CS=0: keep the slab kernel (A/B measurements)
  // rows too long for the layout's count fields are peeled off it (spmv_cs.hpp CsView::peel) and done over the plain CSR
  DevBuf<unsigned> peel_mask;
  DevBuf<int4> peel_blk;
  int npeel = 0, npeel_long = 0;
  long peel_nnz = 0;  // nonzeros in the peeled rows
  // host: mark rows longer than `thresh`; one row block {row, row + 1, first nonzero, end} each.  false: nothing to peel
  bool make_peel(const int *rp_host, int thresh, hipStream_t s) {
    peel_mask.release(); peel_blk.release(); npeel = 0; npeel_long = 0; peel_nnz = 0;
    if (!opts().cs_peel) return false;  // (labs) A/B: reject such patterns as round 1 did
    std::vector<int4> blk;
    std::vector<unsigned> mask;
    for (int r = 0; r < rows; ++r)
      if (rp_host[r + 1] - rp_host[r] > thresh) {
        if (mask.empty()) mask.assign(((size_t)rows + 31) / 32, 0u);
        mask[r >> 5] |= 1u << (r & 31);
        blk.push_back(int4{r, r + 1, rp_host[r], rp_host[r + 1]});
      }
    if (blk.empty()) return false;
    // the longest rows first (those > kPeelLongRow get a whole workgroup each in k_spmv_peeled; starting the long ones early
    // keeps the tail of the launch short), ties in row order: a fixed order, so the reduction partials are deterministic
    std::stable_sort(blk.begin(), blk.end(), [](const int4 &a, const int4 &b) { return a.w - a.z > b.w - b.z; });
    npeel_long = 0;
    for (const int4 &b : blk) { npeel_long += (b.w - b.z > kPeelLongRow) ? 1 : 0; peel_nnz += b.w - b.z; }
    npeel = (int)blk.size();
    peel_mask.upload(mask.data(), mask.size(), s);
    peel_blk.upload(blk.data(), blk.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    return true;
  }
  // ---- virtual rows (spmv_cs.hpp CsView::Rr): long rows cut into pieces that ride in the passes ----
  using VirtPlan = CsVirtPlan;
  static bool virt_enabled() { return opts().cs_virt; }  // (labs) SCS_HIP_CS_VIRT=0: long rows go to the CSR-stream side launch whole (round 2)
  // rows longer than max(lp, what a count field holds) nonzeros -> ceil(len / lp) pieces (rows a field holds stay whole and keep
  // the oracle's summation order; a piece's run is added by ONE lane, so pieces are short whatever the field would hold);
  // fills the peel mask / row blocks {row, row + 1, first piece, end}
  bool plan_virtual(const int *rp, int lp, VirtPlan &P, hipStream_t s) {
    clear_peel();
    if (!cs_plan_virtual(rp, rows, lp, std::max(lp, peel_threshold(1)), P)) return false;
    npeel = (int)P.blk.size();
    npeel_long = 0;  // (a row's pieces are few: one wavefront adds them)
    peel_nnz = P.long_nnz;
    peel_mask.upload(P.mask.data(), P.mask.size(), s);
    peel_blk.upload(P.blk.data(), P.blk.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    return true;
  }
  void adopt_virtual(const VirtPlan &P, hipStream_t s) {
    cs.rows = rows;
    cs.Rr = P.Rr; cs.Rp = P.Rp; cs.npieces = P.V;
    cs.tpart.alloc_zero((size_t)P.V, s);
  }
  bool build_virtual_dev(const DeviceCsr &T, const int *rp, int lp, hipStream_t s) {
    VirtPlan P;
    if (!plan_virtual(rp, lp, P, s)) return false;
    DevBuf<int2> d_info;
    DevBuf<int> vslot;
    d_info.upload(P.rowinfo.data(), P.rowinfo.size(), s);
    vslot.alloc((size_t)nnz);
    hipLaunchKernelGGL(k_cs_vslot, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, T.rowptr.p, T.col.p, cols, (long)nnz, rowptr.p, col.p,
                       d_info.p, P.Rr, P.Rp, P.R, vslot.p);
    HIP_CHECK(hipStreamSynchronize(s));  // (P.rowinfo is read by the upload)
    const bool built = cs.build_from_transpose(P.nchunks * P.R, cols, T.rowptr.p, vslot.p, T.val.p, nnz, s, 1, nullptr, P.R, P.rpt);
    if (opts().debug & DBG_SETUP)
      std::fprintf(stderr, "[scs-hip] column-sorted layout %d x %d: rows longer than %d in pieces of <= %d (%d rows, %ld of %ld nonzeros, %d pieces; chunks of %d + %d slots, %d rows per lane): %s\n",
                   rows, cols, std::max(lp, peel_threshold(1)), lp, npeel, peel_nnz, (long)nnz, P.V, P.Rr, P.Rp, P.rpt, built ? "built" : "a count field overflowed");
    if (!built) { clear_peel(); return false; }
    adopt_virtual(P, s);
    return true;
  }
  bool build_virtual_host(const int *rp, const int *ci, const double *v, int lp, hipStream_t s, HostCs &h) {
    VirtPlan P;
    if (!plan_virtual(rp, lp, P, s)) return false;
    if (!build_cs_virtual(rp, ci, v, rows, cols, P, h)) { clear_peel(); return false; }
    virt_host_plan = P;
    return true;
  }
  VirtPlan virt_host_plan;
  // pieces per pass to aim for (x the passes a chunk is expected to have = the piece length): smaller pieces, more slots
  std::vector<int> virt_piece_lengths() const {
    const long npass_est = std::max<long>(1, (long)nnz / kCsTargetWgs / kCsPass);
    std::vector<int> out;
    for (int per_pass : {24, 12, 6}) out.push_back((int)std::min<long>(per_pass * npass_est, 1L << 20));
    return out;
  }
  static std::vector<int> peel_ladder() {  // (labs) SCS_HIP_CS_PEEL_LADDER=0: rows longer than a count field at once (round 2)
    if (!opts().cs_peel_ladder) return {1};
    return {32, 16, 8, 4, 2, 1};
  }
  int peel_threshold(int split) const {
    int R, rpt;
    cs_pick_geometry(rows, R, rpt, split);
    return cs_peel_threshold(rpt);
  }
  DevBuf<double> cs_part0, cs_part1;  // cs.split == 2 without the in-kernel combine: partial row sums (spmv.hpp EpiPartial / EpiGp::split)
  static bool cs_split_enabled() { return opts().cs_split; }  // SCS_HIP_CS_SPLIT=0: one workgroup per row chunk everywhere (bit-exact sequential row sums; A/B)
  // Workgroups per row chunk.  kind: 0 = A (y-space products), 1 = A' (x-space products), 2 = P.  Taller chunks mean more
  // nonzeros per 128-byte line of the gather vector, i.e. fewer lines per gather instruction — the quantity that bounds
  // these kernels — at the price of partial row sums.  Default: only A' is split, in two, and hands its two partial
  // vectors to the CG update (EpiGp::split: Gp is linear in them) or to k_epi_finish — no combine pass.
  // SCS_HIP_CS_COMBINE=1 (braided kernel only): the partial sums of up to 4 parts are added INSIDE the kernel by the
  // last workgroup of a chunk to arrive, so every product — A too — may be split (SCS_HIP_CS_SPLIT_A / _AT / _P).
  // Measured at the bench size (tools/cs_lab.hip): the 48 MB of partial-sum traffic and the 16-rows-per-lane row sums
  // eat the gather gain (A: 91.5 us unsplit, 95 us split in two + combine; A': 93 us two partial vectors, 100 us four
  // parts + combine) => off by default.
  static bool cs_combine_enabled() {  // (labs)
    return opts().cs_combine && cs_schedule() >= 2;  // (round 5: the round-4 schedule too — k_spmv_cs_il<.., 6> carries the same combine code)
  }
  int cs_pick_split(int kind) const {
    if (!cs_split_enabled() || opts().cs_rpt > 0) return 1;
    if (!cs_combine_enabled()) {
      if (kind != 1) return 1;
      int R, rpt;
      cs_pick_geometry(rows, R, rpt, 2);
      return rpt <= 8 ? 2 : 1;
    }
    { const int v = kind == 0 ? opts().cs_split_a : kind == 1 ? opts().cs_split_at : opts().cs_split_p; if (v == 1 || v == 2 || v == 4) return v; }
    for (int sp : {4, 2}) {
      int R, rpt;
      cs_pick_geometry(rows, R, rpt, sp);
      if ((long)R * (kCsTargetWgs / sp) >= rows && rpt <= 16 && R >= 64 * sp) return sp;  // the chunks still cover all rows in one wave of workgroups
    }
    return 1;
  }
  void cs_after_build(hipStream_t s) {
    cs_part0.release(); cs_part1.release();
    if (!cs.ok || cs.split <= 1) return;
    if (cs_combine_enabled()) cs.enable_combine(s);
    else { cs_part0.alloc_zero((size_t)rows, s); cs_part1.alloc_zero((size_t)rows, s); }
  }
  // T = this matrix transposed (device CSR with the CURRENT values); host: build from this matrix's own host arrays.
  bool build_cs_dev(const DeviceCsr &T, hipStream_t s, int kind) {
    cs.release();
    peel_mask.release(); peel_blk.release(); npeel = 0; npeel_long = 0;
    if (!cs_enabled() || !cs_wanted(rows, cols, nnz) || !opts().slab) return false;
    bool ok = false;
    const int sp = cs_pick_split(kind);
    std::vector<int> rp((size_t)rows + 1);  // row lengths decide what is peeled (O(rows) at init)
    rowptr.download(rp.data(), rp.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    // For every split candidate: first WITHOUT peeling — what the count fields limit is a row's nonzeros inside ONE
    // pass, and a long row whose columns are spread out (a uniformly denser matrix: 100 nonzeros per row over 1e6
    // columns) has ~1 per pass — then, if a count overflowed, with the rows longer than a count field peeled off; and
    // a layout whose peeled rows hold most of the nonzeros is not kept (the side launch would be the product).
    auto attempt = [&](int split) {
      clear_peel();
      if (cs.build_from_transpose(rows, cols, T.rowptr.p, T.col.p, T.val.p, nnz, s, split, nullptr)) return true;
      // the long rows cut into pieces that ride in the passes (one workgroup per chunk: split 1) ...
      if (virt_enabled())
        for (int lp : virt_piece_lengths())
          if (build_virtual_dev(T, rp.data(), lp, s)) return true;
      // ... or, failing that, peeled as FEW rows as the count fields allow: a row of 500 nonzeros has ~50 in each of its chunk's ten passes and
      // rides in them (its gathers share lines with the other rows' there); thresholds from 32 x the field down to the field
      for (int mult : peel_ladder()) {
        if (!make_peel(rp.data(), peel_threshold(split) * mult, s)) continue;  // (no row that long: next rung)
        if (peel_nnz > (nnz / 5) * 3) { clear_peel(); return false; }
        const bool built = cs.build_from_transpose(rows, cols, T.rowptr.p, T.col.p, T.val.p, nnz, s, split, peel_mask.p);
        if (opts().debug & DBG_SETUP)
          std::fprintf(stderr, "[scs-hip] column-sorted layout %d x %d, split %d: rows longer than %d peeled (%d rows, %ld of %ld nonzeros): %s\n",
                       rows, cols, split, peel_threshold(split) * mult, npeel, peel_nnz, (long)nnz, built ? "built" : "a count field overflowed");
        if (built) return true;
      }
      clear_peel();
      return false;
    };
    if (sp > 1) ok = attempt(sp);
    if (!ok) ok = attempt(1);
    if (!ok) clear_peel();
    cs_after_build(s);
    return ok;
  }
  void clear_peel() { peel_mask.release(); peel_blk.release(); npeel = 0; npeel_long = 0; peel_nnz = 0; }
  bool build_cs_host(const int *rp, const int *ci, const double *v, hipStream_t s, int kind) {
    cs.release();
    peel_mask.release(); peel_blk.release(); npeel = 0; npeel_long = 0;
    if (!cs_enabled() || !cs_wanted(rows, cols, nnz) || !opts().slab) return false;
    HostCs h;
    bool ok = false, virt_host = false;
    const int sp = cs_pick_split(kind);
    auto attempt = [&](int split) {  // same policy as build_cs_dev: unpeeled first, then the long rows peeled, capped
      clear_peel();
      if (build_cs(rp, ci, v, rows, cols, h, 0, split, nullptr)) return true;
      if (virt_enabled())
        for (int lp : virt_piece_lengths())
          if (build_virtual_host(rp, ci, v, lp, s, h)) { virt_host = true; return true; }
      for (int mult : peel_ladder()) {
        const int thresh = peel_threshold(split) * mult;
        if (!make_peel(rp, thresh, s)) continue;
        if (peel_nnz > (nnz / 5) * 3) { clear_peel(); return false; }
        std::vector<unsigned> mk(((size_t)rows + 31) / 32, 0u);
        for (int r = 0; r < rows; ++r)
          if (rp[r + 1] - rp[r] > thresh) mk[r >> 5] |= 1u << (r & 31);
        if (build_cs(rp, ci, v, rows, cols, h, 0, split, mk.data())) return true;
      }
      clear_peel();
      return false;
    };
    if (sp > 1) ok = attempt(sp);
    if (!ok) ok = attempt(1);
    if (!ok) { clear_peel(); return false; }
    cs.from_host(h, s);
    if (virt_host) adopt_virtual(virt_host_plan, s);
    cs_after_build(s);
    return true;
  }
  static bool host_setup() { return opts().host_setup; }  // SCS_HIP_SETUP=host: transposition and slab construction on the host (fallback / A-B / tests)
  void set_rowblocks(const int *rp_host, hipStream_t s) {
    std::vector<int4> rb = build_rowblocks(rp_host, rows);
    nblk = (int)rb.size();
    rowblk.upload(rb.data(), rb.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));  // rb is a local
  }
  void upload(int rows_, int cols_, const int *rp, const int *ci, const double *v, hipStream_t s, bool allow_slab = true) {
    rows = rows_; cols = cols_; nnz = rp[rows_];
    rowptr.upload(rp, rows + 1, s);
    col.upload(ci, nnz, s);
    val.upload(v, nnz, s);
    set_rowblocks(rp, s);
    has_slab = false;
    if (allow_slab && slab_wanted(rows, cols) && opts().slab) {  // SCS_HIP_SLAB=0 forces the plain CSR-stream kernel (A/B measurements)
      if (!host_setup()) {
        build_slab_dev(s);
      } else {
        HostSlab hs;
        std::vector<int> src;
        if (build_slab(rp, ci, v, rows, cols, hs, &src)) {
          s_perm.upload(src.data(), src.size(), s);
          s_segptr.upload(hs.segptr.data(), hs.segptr.size(), s);
          s_roff.upload(hs.roff.data(), hs.roff.size(), s);
          s_col.upload(hs.col.data(), hs.col.size(), s);
          s_val.upload(hs.val.data(), hs.val.size(), s);
          s_nchunks = hs.nchunks; s_S = hs.S; s_R = hs.R; s_max_seg = hs.max_seg;
          has_slab = true;
          HIP_CHECK(hipStreamSynchronize(s));  // hs is a local
        }
      }
    }
    HIP_CHECK(hipStreamSynchronize(s));
  }
  // this = src' on the device (setup_dev.hpp).  false: a row is too long for the one-lane sort (caller falls back).
  bool transpose_from(const DeviceCsr &src, hipStream_t s) {
    rows = src.cols; cols = src.rows; nnz = src.nnz;
    rowptr.alloc_zero((size_t)rows + 1, s);
    col.alloc_zero((size_t)std::max(nnz, 1L), s);
    val.alloc_zero((size_t)std::max(nnz, 1L), s);
    DevBuf<int> cursor, perm, tmp, flag;
    cursor.alloc_zero((size_t)rows + 1, s);
    perm.alloc_zero((size_t)std::max(nnz, 1L), s);
    tmp.alloc_zero((size_t)(rows / kScanTile + 4), s);
    flag.alloc_zero(1, s);
    if (nnz > 0) hipLaunchKernelGGL(k_count_index, dim3(vec_blocks(nnz)), dim3(kVecThreads), 0, s, src.col.p, nnz, cursor.p);
    device_exclusive_scan(cursor.p, rowptr.p, rows, tmp.p, s);
    HIP_CHECK(hipMemcpyAsync(cursor.p, rowptr.p, sizeof(int) * rows, hipMemcpyDeviceToDevice, s));
    // small matrices: a wavefront per row (setup_dev.hpp; the same result, a shorter link in the dispatch chain of a small scs_init)
    const bool per_wave = std::max(rows, src.rows) <= kTransposeWaveRows;
    if (per_wave) {
      const int wpb = kVecThreads / 64;
      hipLaunchKernelGGL(k_transpose_scatter_w, dim3(std::max(1, std::min(ceil_div(src.rows, wpb), kMaxVecBlocks))), dim3(kVecThreads), 0, s, src.rowptr.p,
                         src.col.p, src.rows, cursor.p, col.p, perm.p);
      hipLaunchKernelGGL(k_sort_rows_w, dim3(std::max(1, std::min(ceil_div(rows, wpb), kMaxVecBlocks))), dim3(kVecThreads), 0, s, rowptr.p, col.p, perm.p, rows,
                         flag.p);
    } else {
      hipLaunchKernelGGL(k_transpose_scatter, dim3(vec_blocks(src.rows)), dim3(kVecThreads), 0, s, src.rowptr.p, src.col.p, src.rows,
                         cursor.p, col.p, perm.p);
      hipLaunchKernelGGL(k_sort_rows, dim3(vec_blocks(rows)), dim3(kVecThreads), 0, s, rowptr.p, col.p, perm.p, rows, flag.p);
    }
    if (nnz > 0) hipLaunchKernelGGL(k_gather_f64, dim3(vec_blocks(nnz)), dim3(kVecThreads), 0, s, val.p, src.val.p, perm.p, nnz);
    int too_long = 0;
    std::vector<int> rp((size_t)rows + 1);
    HIP_CHECK(hipMemcpyAsync(&too_long, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    rowptr.download(rp.data(), rp.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    if (too_long) return false;
    set_rowblocks(rp.data(), s);
    has_slab = false;
    return true;
  }
  // L2-blocked copy of the CURRENT csr arrays, built on the device (same layout as spmv.hpp build_slab)
  void build_slab_dev(hipStream_t s) {
    has_slab = false;
    if (!slab_wanted(rows, cols) || !opts().slab) return;
    SlabGeom g;
    g.rows = rows; g.cols = cols; g.R = slab_pick_rows(rows); g.shift = slab_shift();
    g.S = (int)(((long)cols + (1L << g.shift) - 1) >> g.shift);
    g.nchunks = (rows + g.R - 1) / g.R;
    const long nseg = (long)g.nchunks * g.S;
    DevBuf<int> seg_size, tmp, flag;
    seg_size.alloc_zero((size_t)nseg + 1, s);
    tmp.alloc_zero((size_t)(nseg / kScanTile + 4), s);
    flag.alloc_zero(1, s);
    s_roff.alloc_zero((size_t)nseg * (g.R + kSlabRoffPad), s);
    s_segptr.alloc_zero((size_t)nseg + 1, s);
    hipLaunchKernelGGL(k_slab_count, dim3(vec_blocks(rows)), dim3(kVecThreads), 0, s, rowptr.p, col.p, g, s_roff.p, flag.p);
    hipLaunchKernelGGL(k_slab_scan, dim3((unsigned)nseg), dim3(kScanThreads), 0, s, g, s_roff.p, seg_size.p, flag.p);
    device_exclusive_scan(seg_size.p, s_segptr.p, nseg, tmp.p, s);
    std::vector<int> sizes((size_t)nseg);
    int total = 0, overflow = 0;
    HIP_CHECK(hipMemcpyAsync(sizes.data(), seg_size.p, sizeof(int) * nseg, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&total, s_segptr.p + nseg, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&overflow, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    long check = 0;
    int max_seg = 0;
    for (int v : sizes) { check += v; max_seg = std::max(max_seg, v); }
    if (overflow || check != (long)total || check > 2000000000L) {  // uint16 offsets or int32 positions do not fit: no slab copy
      s_roff.release(); s_segptr.release();
      return;
    }
    s_col.alloc_zero((size_t)std::max(total, 1), s);
    s_val.alloc_zero((size_t)std::max(total, 1), s);
    hipLaunchKernelGGL(k_slab_fill, dim3(vec_blocks(rows)), dim3(kVecThreads), 0, s, rowptr.p, col.p, val.p, g, s_roff.p, s_segptr.p,
                       s_col.p, s_val.p);
    hipLaunchKernelGGL(k_slab_pad, dim3(vec_blocks(nseg)), dim3(kVecThreads), 0, s, g, s_roff.p, s_segptr.p, s_col.p, s_val.p);
    HIP_CHECK(hipStreamSynchronize(s));
    s_nchunks = g.nchunks; s_S = g.S; s_R = g.R; s_max_seg = max_seg;
    has_slab = true;
  }
  SpmvMat view() const {
    SpmvMat M;
    M.csr = CsrView{rowptr.p, col.p, val.p, rowblk.p, rows, cols, nblk, nnz};
    M.use_slab = has_slab;
    if (has_slab) M.slab = SlabView{s_segptr.p, s_roff.p, s_col.p, s_val.p, rows, cols, s_nchunks, s_S, s_R, s_max_seg};
    M.use_cs = cs.ok;
    if (cs.ok) {
      M.cs = cs.view(); M.part0 = cs_part0.p; M.part1 = cs_part1.p;
      M.cs.peel = npeel > 0 ? peel_mask.p : nullptr;
      M.peel_blk = peel_blk.p;
      M.npeel = npeel;
      M.nlong = npeel_long;
    }
    return M;
  }
  int nwg() const { return cs.ok ? (cs.combine() ? cs.nchunks : cs.nchunks * cs.split) + peel_wgs_for(npeel, npeel_long) : has_slab ? s_nchunks : nblk; }
  // after the CSR values were rescaled on the device: refresh the slab copy and drop the index map
  void refresh_slab(hipStream_t s, bool drop_perm) {
    if (!has_slab || s_perm.n == 0) return;  // (device-built slabs are made from the already equilibrated values)
    const long cnt = (long)s_val.n;
    hipLaunchKernelGGL(k_gather_vals, dim3(vec_blocks(cnt)), dim3(kVecThreads), 0, s, s_val.p, val.p, s_perm.p, cnt);
    if (drop_perm) {
      HIP_CHECK(hipStreamSynchronize(s));
      s_perm.release();
    }
  }
};

// K12 on the device: equilibrate the three resident layouts in place; D (m) and E (n) accumulate the scalings.
static void device_normalize(DeviceCsr &At, DeviceCsr &Ar, DeviceCsr *Pf, const HostCone &cone, DevBuf<double> &D,
                             DevBuf<double> &E, hipStream_t s) {
  const int m = Ar.rows, n = At.rows;
  DevBuf<double> Dt, Et, Ep;
  Dt.alloc(m);
  Et.alloc(n);
  if (Pf) Ep.alloc(n);
  D.alloc(m);
  E.alloc(n);
  hipLaunchKernelGGL(k_fill, dim3(vec_blocks(m)), dim3(kVecThreads), 0, s, D.p, 1.0, (long)m);
  hipLaunchKernelGGL(k_fill, dim3(vec_blocks(n)), dim3(kVecThreads), 0, s, E.p, 1.0, (long)n);
  // non-separable cone blocks (everything after the z/l/box rows)
  // (every block behind the separable rows, the one-row ones too: normalize_dev.hpp k_pass_finish)
  std::vector<int> boff, blen;
  const int prefix = (int)std::min<long>(cone.boundaries[0], m);
  long count = cone.boundaries[0];
  for (size_t i = 1; i < cone.boundaries.size(); ++i) {
    if (cone.boundaries[i] >= 1) { boff.push_back((int)count); blen.push_back(cone.boundaries[i]); }
    count += cone.boundaries[i];
  }
  const bool fused_finish = opts().norm_fuse;  // (labs) SCS_HIP_NORM_FUSE=0: the four launches of rounds 1-4 (A/B; same bits)
  DevBuf<int> dboff, dblen;
  const int nblocks = (int)boff.size();
  if (nblocks) { dboff.upload(boff.data(), boff.size(), s); dblen.upload(blen.data(), blen.size(), s); }
  // norms of pass p+1 come out of the rescale sweep of pass p (k_rescale_norm); the very first norms need their own sweep
  DevBuf<double> Dn, En;
  Dn.alloc(m);
  En.alloc(n);
  auto sweep = [&](DeviceCsr &M, const double *rs, const double *cs, int l2, double *out) {
    if (M.nblk > 0)
      hipLaunchKernelGGL(k_rescale_norm, dim3(M.nblk), dim3(kSpmvThreads), 0, s, M.view().csr, M.val.p, rs, cs, l2, out);
  };
  // (the sweeps of a pass are independent of each other: one launch for all of them, normalize_dev.hpp k_rescale_norm3)
  auto sweeps = [&](const double *Dfac, const double *Efac, int l2, double *Dout, double *Eout, double *Pout) {
    if (!fused_finish) {
      sweep(Ar, Dfac, Efac, l2, Dout);
      sweep(At, Efac, Dfac, l2, Eout);
      if (Pf) sweep(*Pf, Efac, Efac, l2, Pout);
      return;
    }
    const int n1 = Ar.nblk, n2 = At.nblk, n3 = Pf ? Pf->nblk : 0;
    if (n1 + n2 + n3 <= 0) return;
    const CsrView v1 = Ar.view().csr, v2 = At.view().csr, v3 = Pf ? Pf->view().csr : v1;
    hipLaunchKernelGGL(k_rescale_norm3, dim3(n1 + n2 + n3), dim3(kSpmvThreads), 0, s, v1, Ar.val.p, Dfac, Efac, Dout, n1, v2, At.val.p, Efac, Dfac, Eout, n2,
                       v3, Pf ? Pf->val.p : (double *)nullptr, Efac, Efac, Pout, l2);
  };
  sweeps(nullptr, nullptr, 0, Dt.p, Et.p, Pf ? Ep.p : nullptr);
  for (int pass = 0; pass < 26; ++pass) {
    const int l2 = pass >= 25 ? 1 : 0;
    const int l2_next = pass + 1 >= 26 ? -1 : (pass + 1 >= 25 ? 1 : 0);
    if (Pf) hipLaunchKernelGGL(k_combine, dim3(vec_blocks(n)), dim3(kVecThreads), 0, s, Et.p, Ep.p, n, l2);
    if (fused_finish) {
      const int nbD = prefix > 0 ? vec_blocks(prefix) : 0, nbE = vec_blocks(n), nbB = ceil_div(nblocks, kVecThreads / 64);
      hipLaunchKernelGGL(k_pass_finish, dim3(nbD + nbE + nbB), dim3(kVecThreads), 0, s, Dt.p, D.p, prefix, Et.p, E.p, n, (const int *)dboff.p,
                         (const int *)dblen.p, nblocks, l2, nbD, nbE);
    } else {
      if (l2) {
        hipLaunchKernelGGL(k_sqrt_inplace, dim3(vec_blocks(m)), dim3(kVecThreads), 0, s, Dt.p, m);
        hipLaunchKernelGGL(k_sqrt_inplace, dim3(vec_blocks(n)), dim3(kVecThreads), 0, s, Et.p, n);
      }
      if (nblocks)
        hipLaunchKernelGGL(k_enforce_blocks, dim3(ceil_div(nblocks, kVecThreads / 64)), dim3(kVecThreads), 0, s, Dt.p, dboff.p,
                           dblen.p, nblocks, l2);
      hipLaunchKernelGGL(k_invsqrt_acc, dim3(vec_blocks(m)), dim3(kVecThreads), 0, s, Dt.p, D.p, m);
      hipLaunchKernelGGL(k_invsqrt_acc, dim3(vec_blocks(n)), dim3(kVecThreads), 0, s, Et.p, E.p, n);
    }
    sweeps(Dt.p, Et.p, l2_next, Dn.p, En.p, Pf ? Ep.p : nullptr);
    std::swap(Dt.p, Dn.p);
    std::swap(Et.p, En.p);
  }
  HIP_CHECK(hipStreamSynchronize(s));  // Dt/Et/Ep and the block arrays are locals
}

// b_hat = sigma D b, c_hat = sigma E c on the device vector h = [c; b]; returns sigma
static double device_normalize_b_c(DevBuf<double> &h, int n, int m, const DevBuf<double> &D, const DevBuf<double> &E,
                                   DevBuf<double> &part, double *h_pin, hipStream_t s) {
  const int nbn = vec_blocks(n), nbm = vec_blocks(m);
  hipLaunchKernelGGL(k_scale_by_vec, dim3(nbn), dim3(kVecThreads), 0, s, h.p, E.p, n, part.p);
  hipLaunchKernelGGL(k_scale_by_vec, dim3(nbm), dim3(kVecThreads), 0, s, h.p + n, D.p, m, part.p + kMaxVecBlocks);
  std::vector<double> pm(2 * kMaxVecBlocks, 0.0);
  HIP_CHECK(hipMemcpyAsync(pm.data(), part.p, sizeof(double) * 2 * kMaxVecBlocks, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  double nc = 0., nb = 0.;
  for (int i = 0; i < nbn; ++i) nc = std::max(nc, pm[i]);
  for (int i = 0; i < nbm; ++i) nb = std::max(nb, pm[kMaxVecBlocks + i]);
  double sigma = std::max(nc, nb);
  sigma = sigma < 1e-4 ? 1.0 : sigma;
  sigma = sigma > 1e4 ? 1e4 : sigma;
  sigma = safediv_pos(1.0, sigma);
  hipLaunchKernelGGL(k_scale_scalar, dim3(vec_blocks((long)n + m)), dim3(kVecThreads), 0, s, h.p, sigma, (long)n + m);
  (void)h_pin;
  return sigma;
}

struct Residuals {
  int last_iter = -1;
  double tau = 0, kap = 0;
  double nm_pri_n = 0, nm_dual_n = 0;  // normalised ||Ax+s-b tau||, ||Px+A'y+c tau||
  double nm_ax_s_btau = 0, nm_ax_s = 0, nm_ax = 0, nm_s = 0;
  double nm_px_aty_ctau = 0, nm_px = 0, nm_aty = 0;
  double bty_tau = 0, ctx_tau = 0, xt_p_x_tau = 0;
  double bty = 0, ctx = 0, xt_p_x = 0, gap = 0, pobj = 0, dobj = 0;
  double res_pri = 0, res_dual = 0, res_infeas = NAN, res_unbdd_a = NAN, res_unbdd_p = NAN;
  // extras for the CSV log (normalised space and 2-norms)
  double sq_pri_n = 0, sq_pri_o = 0, sq_dual_n = 0, sq_dual_o = 0, nm_ax_s_n = 0, nm_px_n = 0, nm_aty_n = 0;
  double bty_tau_n = 0, ctx_tau_n = 0, xt_p_x_tau_n = 0, kap_n = 0;
};

}  // namespace scship

using namespace scship;

static void write_csv_row(FILE *f, int iter, const Residuals &r, double scale, const double *diffs, double aa_norm,
                          double time_s);

// A spinning multi-workgroup kernel (k_psd_sweep_mc, k_cg_persist) gave up at a barrier: scs_solve restarts the solve without them
struct SpinTimeout : std::runtime_error {
  using std::runtime_error::runtime_error;
};
static std::atomic<long> g_spin_fallbacks{0};  // scs_hip_spin_fallbacks(): tests
// Spinning kernels need ALL their workgroups on the device at once.  One workspace alone sizes its grid for that; two workspaces of a
// device that launch such grids on different streams at the same time (threads with their own SCS objects: R:test/test_thread_safety.py:78-93)
// could each get half of theirs placed and wait for the other half for good.  So inside a process the spinning launches of a device
// form a chain: once more than one workspace of the device uses them, each launch waits for the event recorded behind the previous one
// (hipStreamWaitEvent: nothing on the host waits) and leaves its own.  A lone user pays nothing.
struct SpinChain {
  std::mutex mu;
  int users = 0;
  hipEvent_t last = nullptr;
  hipStream_t last_stream = nullptr;
};
static SpinChain &spin_chain(int device) {
  static SpinChain c[64];
  return c[device & 63];
}

// ============================================================== workspace
struct ScsHipWork {
  // first member = destroyed last: ends the window in which this workspace's device blocks go to the block pool (common.hpp DevPool)
  struct PoolWindowEnd {
    bool armed = false;
    ~PoolWindowEnd() { if (armed) --t_pool_release; }
  } pool_window_end;
  std::unique_ptr<Arena> arena;  // small problems: all device buffers of the workspace come from here (FIRST member: destroyed last)
  int device = 0;  // the HIP device this workspace (stream, buffers, events) lives on
  int n = 0, m = 0;
  long l = 0;
  ScsSettings stgs{};
  double scale = 0.1;
  HostCone cone;
  HostScaling scal;
  bool normalized = false, has_P = false;
  std::vector<double> b_orig, c_orig;
  double nm_b_orig = 0, nm_c_orig = 0;
  double setup_time = 0;
  std::string log_csv_filename, write_data_filename;

  hipStream_t stream = nullptr;
  bool owns_stream = true, pooled_stream = false, stream_shared = false;
  void *pinned_block = nullptr;  // all pinned host scalars / flags of the workspace (g_pinned)
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  double *h_pin = nullptr;  // pinned scalars
  int *h_flags = nullptr;   // pinned flags
  double *h_params = nullptr, *d_params = nullptr;  // mapped pinned per-iteration scalars (P_*), slot in use (2 slots)
  double *h_params_base = nullptr, *d_params_base = nullptr;
  // run-ahead mode (see F_STALL in vec.hpp): plain iterations are enqueued whole and one ahead of the host's view
  bool pipelined = false;
  int pipe_chunk_override = 0, pipe_stalls = 0;  // tests: SCS_HIP_PIPELINE=N forces CG chunks of N steps (=> stalls)
  const int *stall = nullptr;      // fl + F_STALL while a run-ahead iteration is being enqueued, else nullptr
  int *stall_fl = nullptr;         // fl (or nullptr): k_tau_dots raises the stall, k_cone_pre parks the CG kernels
  int *h_flags_slot[2] = {nullptr, nullptr};
  hipEvent_t ev_iter[2] = {nullptr, nullptr};
  hipEvent_t ev_prof[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // in-situ K1/K2 samples of the run-ahead loop
  int prof_step[2] = {-1, -1};  // CG step (0-based) bracketed by ev_prof[slot], -1 = none
  hipEvent_t ev_cone[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // in-situ: the cone kernels of a queued iteration
  bool cone_sampled[2] = {false, false};
  double prof_cone_ms = 0;
  long prof_cone_n = 0;

  // hipGraphs of the launch-bound inner loop (built lazily at the first solve):
  //   g_pre[i] : iterate normalisation, rhs, CG start + kGraphSteps[i] CG steps + flag read-back
  //   g_cg[i]  : kGraphSteps[i] further CG steps + flag read-back
  //   g_post   : y recovery, tau, cone projections, dual update (iterations without a convergence check)
  static constexpr int kNumGraphs = 5;
  const int kGraphSteps[kNumGraphs] = {1, 2, 4, 8, 16};
  hipGraphExec_t g_pre[kNumGraphs] = {}, g_cg[kNumGraphs] = {}, g_post = nullptr;
  bool graphs_ready = false;
#ifdef SCS_HIP_LABS
  bool graphs_enabled = true;
#else
  static constexpr bool graphs_enabled = false;  // (hipGraph replay lives in the labs build: 5 % slower than eager launches at config 2)
#endif
  // small problems: the whole PCG solve of an iteration is one persistent launch (cg_persist.hpp)
#ifdef SCS_HIP_LABS
  int persist_wgs = 0, persist_ng = 1;  // 0 = launch-per-kernel path
  DevBuf<unsigned> persist_bar;
#else
  static constexpr int persist_wgs = 0, persist_ng = 1;  // (the persistent kernel lives in the labs build: never faster than launch-per-kernel)
#endif

  DeviceCsr At;  // CSR(A') == caller's CSC(A): rows n, cols m   (x-space outputs)
  DeviceCsr Ar;  // CSR(A): rows m, cols n                        (y-space outputs)
  DeviceCsr Pf;  // full symmetric CSR(P)
  DevBuf<double> Pdiag;

  DevBuf<double> v, v_prev, u, ut, rsk, g, h, diag_r, D, E, Dinv, Einv;
  DevBuf<double> cg_b, cg_p, cg_r, cg_Gp, cg_M, tmp_m, ws, px;
  DevBuf<double> part, part2, sc, out;  // part2: partials of k_cg_update (read by k_cg_dir while `part` is reused), of k_prep
  DevBuf<double> part_v;                // sum-of-squares partials of v for the next k_prep
  bool v_norm_fresh = false;
  DevBuf<int> fl;
  DevBuf<double> solx, soly, sols;
  bool sol_on_device = false;  // solx/soly/sols hold the final (x, y, s) of the last solve
  // large solutions leave through a pinned mirror owned by the workspace (the caller's arrays are never handed to the runtime,
  // see scs_hip_runtime_env): three DMA copies in flight, each array moved on by a few host threads as soon as it has landed
  double *sol_pin = nullptr;
  bool sol_pin_refused = false;
  hipEvent_t sol_ev[3] = {nullptr, nullptr, nullptr};
  static constexpr size_t kSolMirrorMin = (size_t)1 << 20;  // bytes of x | y | s from which the mirror is used
  static void spread_memcpy(void *dst, const void *src, size_t bytes) {
    const int nt = bytes >= ((size_t)8 << 20) ? 4 : 1;
    if (nt == 1) { std::memcpy(dst, src, bytes); return; }
    const size_t part = (bytes / nt + 4095) & ~(size_t)4095;
    std::thread th[3];
    for (int t = 1; t < nt; ++t) {
      const size_t o = std::min(bytes, part * t), c = std::min(bytes - o, part);
      th[t - 1] = std::thread([=] { if (c) std::memcpy((char *)dst + o, (const char *)src + o, c); });
    }
    std::memcpy(dst, src, std::min(bytes, part));
    for (int t = 1; t < nt; ++t) th[t - 1].join();
  }
  void ensure_solution_mirror() {  // (scs_init calls this: pinning 40 MB costs milliseconds)
    const size_t bytes = sizeof(double) * ((size_t)n + 2 * (size_t)m);
    if (sol_pin || sol_pin_refused || bytes < kSolMirrorMin) return;
    if (hipHostMalloc((void **)&sol_pin, bytes, hipHostMallocDefault) != hipSuccess) {  // (no pinned memory left: the runtime's own staging)
      (void)hipGetLastError();
      sol_pin = nullptr;
      sol_pin_refused = true;
      return;
    }
    for (auto &e : sol_ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  void download_solution(ScsSolution *sol) {
    ensure_solution_mirror();
    if (!sol_pin) {
      solx.download(sol->x, n, stream);
      soly.download(sol->y, m, stream);
      sols.download(sol->s, m, stream);
      HIP_CHECK(hipStreamSynchronize(stream));
      return;
    }
    double *hx = sol_pin, *hy = sol_pin + n, *hs = sol_pin + n + m;
    solx.download(hx, n, stream); HIP_CHECK(hipEventRecord(sol_ev[0], stream));
    soly.download(hy, m, stream); HIP_CHECK(hipEventRecord(sol_ev[1], stream));
    sols.download(hs, m, stream); HIP_CHECK(hipEventRecord(sol_ev[2], stream));
    HIP_CHECK(hipEventSynchronize(sol_ev[0])); spread_memcpy(sol->x, hx, sizeof(double) * n);
    HIP_CHECK(hipEventSynchronize(sol_ev[1])); spread_memcpy(sol->y, hy, sizeof(double) * m);
    HIP_CHECK(hipEventSynchronize(sol_ev[2])); spread_memcpy(sol->s, hs, sizeof(double) * m);
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  int part_len = 0;

  // cones
  DevBuf<int> soc_off, soc_dim, soc_big;
  int n_soc = 0, n_soc_big = 0, soc_G = 64;  // soc_G: lanes per cone in k_proj_soc_wave (cones.hpp soc_group)
  DevBuf<double> pow_a, box_bl, box_bu;
  DevBuf<double> box_bl_orig, box_bu_orig, box_parts;
  DevBuf<unsigned> cg_ticket;  // k_cg_update_dir's arrival counter (0 between launches)
  DevBuf<unsigned> box_ticket;  // the caller's bounds (the working copies follow the row scaling): footer diagnostics
  DevBuf<int> psd_off, psd_order;    // orders > kPsdSmallMax first (n_psd_big of them), then the small ones
  DevBuf<long> psd_woff;
  std::vector<long> psd_woff_h;  // host copies (scs_hip_psd_refine_stats)
  std::vector<int> psd_order_h;
  DevBuf<double> psd_scratch;
  int n_psd = 0, n_psd_big = 0;
  // split mode of the block kernel (psd.hpp): worth it when the large matrices alone leave most CUs idle
  static constexpr int kPsdSplitRounds = 2;  // x kPsdLogSweeps = 6 sweeps: a cold start needs ~9; the LAST round is the one behind the refinement stage
  bool psd_split = false;
  int psd_max_np = 0, psd_max_tiles = 0;
  // complex PSD cones: projected through the packed 2k x 2k real embedding held in cs_stage (psd.hpp)
  DevBuf<int> cs_off, cs_order, cs_poff, cs_porder;  // same ordering: embeddings of order > kPsdSmallMax first
  DevBuf<long> cs_soff, cs_woff;
  DevBuf<double> cs_stage;
  int n_cs = 0, n_cs_big = 0;

  // batched PSD projection of `count` packed matrices (the first `big` of order > kPsdSmallMax): K9 + its one-wave variant
  void launch_psd(double *base, const int *off, const int *order, const long *woff, int count, int big) {
    if (big > 0) {
      PsdBatch B{off, order, woff, big};
      if (psd_split) {
        // few large matrices: sweeps (A only) -> V updates over 16-row strips on the idle CUs -> reconstruction
        const int gper = psd_gemm_tasks(std::max(psd_max_tiles, 1));  // tasks (= workgroups of one wavefront) per matrix, dealt to the XCDs in runs
        const dim3 gg(psd_gemm_grid(gper, big)), gb(kPsdGemmThreads);
        const dim3 gt(psd_xcd_grid(std::max(psd_max_tiles, 1), big));
        // front: unpack, V = I / V' on many CUs; orders 0 / 1 and the periodic re-orthogonalisation of V in the one-workgroup kernel
        hipLaunchKernelGGL(k_psd_front, gt, dim3(kPsdFrontThreads), 0, stream,
                           (const double *)base, B, psd_scratch.p, psd_warm, stall);
        hipLaunchKernelGGL(k_proj_psd<3>, dim3(big), dim3(kPsdThreads), kPsdLdsBytes, stream, base, B, psd_scratch.p, psd_warm, 0, stall, psd_tol2,
                           psd_refine_default(false), 0);
        hipLaunchKernelGGL(k_psd_gemm<PSD_G1>, gg, gb, 0, stream, base, B, psd_scratch.p, psd_warm, stall, gper);
        hipLaunchKernelGGL(k_psd_gemm<PSD_G2>, gg, gb, 0, stream, base, B, psd_scratch.p, psd_warm, stall, gper);
        int mc = (in_capture || !fl.p) ? 1 : psd_mc_members(big);  // (the multi-CU kernel polls the workspace's error flag at its barriers)
        PsdRefineCfg R = psd_refine;
        if ((size_t)32 * psd_max_np * sizeof(double) > 160 * 1024) R.on = 0;  // k_psd_apply_q keeps two 16-row strips in LDS
        std::unique_ptr<SpinLink> link;
        if (mc > 1) link.reset(new SpinLink(this));  // spinning launches of this device, one grid at a time (SpinChain)
        for (int round = 0; round < kPsdSplitRounds; ++round) {
          const int post = (R.on && round == kPsdSplitRounds - 1) ? 1 : 0;
          if (post) {
            // the refinement stage (psd.hpp psd_stop_test): matrices the sweeps left REFINABLE get the mixed-sign part of S = V'AV
            // removed by GEMMs; the round behind it re-tests them (and goes on sweeping whatever is not done: nothing is lost)
            hipLaunchKernelGGL(k_psd_gemm<PSD_COMM>, gg, gb, 0, stream, base, B, psd_scratch.p, psd_warm, stall, gper);
            hipLaunchKernelGGL(k_psd_gemm<PSD_KK>, gg, gb, 0, stream, base, B, psd_scratch.p, psd_warm, stall, gper);
            hipLaunchKernelGGL(k_psd_gemm<PSD_T>, gg, gb, 0, stream, base, B, psd_scratch.p, psd_warm, stall, gper);
            hipLaunchKernelGGL(k_psd_gemm<PSD_S1>, gg, gb, 0, stream, base, B, psd_scratch.p, psd_warm, stall, gper);
            hipLaunchKernelGGL(k_psd_apply_q, gt, dim3(kPsdApplyThreads), (size_t)32 * psd_max_np * sizeof(double), stream, B,
                               psd_scratch.p, stall);
          }
          if (mc > 1) {  // sweeps of one matrix over `mc` CUs (k_psd_sweep_mc): cooperative launch, spinning barriers
            double *scr = psd_scratch.p;
            int G = mc, rnd = round;
            int *err = fl.p + F_PERSIST_ERR;
            const int *st = stall;
            int la = psd_mc_look_ahead;
            const double *tl = psd_tol2;
            PsdRefineCfg Rr = R;
            int pst = post;
            long budget = spin_budget;
            void *args[] = {&B, &scr, &rnd, &G, &la, &err, &st, &tl, &Rr, &pst, &budget};
            if (psd_mc_coop) {
              const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_psd_sweep_mc), dim3((unsigned)psd_mc_grid(big, mc)),
                                                             dim3(kPsdThreads), args, (unsigned)kPsdMcLdsBytes, stream);
              if (e != hipSuccess) {  // the runtime cannot co-schedule the grid (it only refuses the FIRST round: nothing ran yet)
                (void)hipGetLastError();
                if (round > 0) HIP_CHECK(e);
                psd_mc_cap = 0;  // from now on: one workgroup per matrix
                mc = 1;
              }
            } else  // SCS_HIP_PSD_COOP=0: ordinary launch (rocprofv3 7.2 crashes at exit after a cooperative launch)
              hipLaunchKernelGGL(k_psd_sweep_mc, dim3((unsigned)psd_mc_grid(big, mc)), dim3(kPsdThreads), kPsdMcLdsBytes, stream, B, scr, rnd, G,
                                 la, err, st, tl, Rr, pst, budget);
          }
          if (mc <= 1)
          hipLaunchKernelGGL(k_proj_psd<1>, dim3(big), dim3(kPsdThreads), kPsdLdsBytes, stream, base, B, psd_scratch.p, psd_warm, round, stall, psd_tol2,
                             R, post);
          hipLaunchKernelGGL(k_psd_apply_v, gt, dim3(kPsdApplyThreads), (size_t)16 * psd_max_np * sizeof(double),
                             stream, B, psd_scratch.p, stall);
        }
        link.reset();
        hipLaunchKernelGGL(k_psd_fmap, gt, dim3(256), 0, stream, B, psd_scratch.p, stall);
        hipLaunchKernelGGL(k_psd_gemm<PSD_R1>, gg, gb, 0, stream, base, B, psd_scratch.p, psd_warm, stall, gper);
        hipLaunchKernelGGL(k_psd_gemm<PSD_R2>, gg, gb, 0, stream, base, B, psd_scratch.p, psd_warm, stall, gper);
      } else {
        hipLaunchKernelGGL(k_proj_psd<0>, dim3(big), dim3(kPsdThreads), kPsdLdsBytes, stream, base, B, psd_scratch.p, psd_warm, 0, stall, psd_tol2,
                           psd_refine_default(false), 0);
      }
    }
    if (count > big) {
      PsdBatch B{off + big, order + big, woff + big, count - big};
      if (psd_small_one_wave)
        hipLaunchKernelGGL(k_proj_psd_small, dim3(count - big), dim3(64), 0, stream, base, B, psd_scratch.p, psd_warm, stall, psd_tol2);
      else
        hipLaunchKernelGGL(k_proj_psd_small4, dim3(count - big), dim3(kPsdSmallThreads), 0, stream, base, B, psd_scratch.p, psd_warm, stall,
                           psd_tol2);
    }
  }
  // Members (CUs) per matrix for the split-mode sweeps: as many as fit when every matrix gets the same number and a
  // group stays inside one XCD (grid = 8 * G * ceil(count / 8) workgroups, all co-resident: cooperative launch).
  // SCS_HIP_PSD_MC=G forces G (0 / 1: the one-workgroup sweep kernel).
  // small matrices (order <= 32): four wavefronts per matrix (psd.hpp d_proj_psd_small4); SCS_HIP_PSD_SMALL_WAVES=1: the one-wavefront kernel (lab; agrees to rounding)
  // SCS_HIP_SOC_PSD_FUSE=0: separate launches for short SOCs and small PSD matrices (same bits)
  bool soc_psd_one_launch = opts().soc_psd_fuse;  // (labs switch)
  // Round 5: GEMM-only refinement of the sign split instead of the last Jacobi sweep(s) in split mode (psd.hpp psd_stop_test).
  // SCS_HIP_PSD_REFINE=0: strict sweeps only (bit-identical to the one-launch kernel); SCS_HIP_PSD_GATE_K / _OFF / _OMEGA: the gate (lab knobs).
  PsdRefineCfg psd_refine = [] {
    const Options &o = opts();
    PsdRefineCfg r = psd_refine_default(o.psd_refine);
    if (o.psd_gate_k > 0.) r.k2 = o.psd_gate_k * o.psd_gate_k;      // (labs: the gate)
    if (o.psd_gate_off > 0.) r.off2 = o.psd_gate_off * o.psd_gate_off;
    if (o.psd_gate_omega > 0.) r.omega = o.psd_gate_omega;
    return r;
  }();
  bool psd_small_one_wave = opts().psd_small_one_wave;  // (labs)
  int psd_mc_look_ahead = opts().psd_la ? 1 : 0;        // (labs switch) one barrier per step
  // Round 4: ORDINARY launch by default.  hipLaunchCooperativeKernel guarantees co-residency of the grid, but on this runtime it costs
  // ~0.1 ms per launch in a fresh process and ~2 ms per launch once the process has driven other workspaces / streams before (config 4 as
  // the second workload of a bench run: 224 iters/s in the steady window and 245 over a whole solve against 462 / 521 with the ordinary
  // launch; cold window 495 vs 522; tools/dbg/c4_after.py, profiles/r04_psd_coop.txt).  The ordinary launch is safe for the same reason the
  // cooperative one is accepted: the grid is sized to fit the device at one workgroup per CU (psd_mc_cap, occupancy query), the
  // dispatcher places workgroups in order, and a kernel of another stream that holds CUs finishes without waiting for this one — a group
  // whose members are late spins within its budget (F_PERSIST_ERR otherwise: an error, not a hang).  SCS_HIP_PSD_COOP=1: cooperative launch.
  bool psd_mc_coop = opts().psd_coop;  // (labs)
  int psd_mc_cap = -1;  // co-resident workgroups of k_psd_sweep_mc on this device (0: no cooperative launch)
  long spin_budget = 1L << opts().spin_budget_log2;  // barrier polls before a member gives up (SCS_HIP_SPIN_BUDGET_LOG2; tests: 0)
  int psd_mc_forced = opts().psd_mc;                 // SCS_HIP_PSD_MC at the workspace's creation (-1: pick)
  bool psd_mc_nocheck = opts().psd_mc_nocheck;       // (labs: tests of the refused launch)
  bool spin_user = false;
  hipEvent_t ev_spin = nullptr;
  void spin_register() {  // before this workspace's first spinning launch
    if (spin_user) return;
    SpinChain &c = spin_chain(device);
    bool others;
    {
      std::lock_guard<std::mutex> lk(c.mu);
      others = ++c.users >= 2;
    }
    spin_user = true;
    HIP_CHECK(hipEventCreateWithFlags(&ev_spin, hipEventDisableTiming));
    if (others) HIP_CHECK(hipDeviceSynchronize());  // what the others launched before they had to leave events is done now
  }
  void spin_unregister() {
    if (!spin_user) return;
    SpinChain &c = spin_chain(device);
    {
      std::lock_guard<std::mutex> lk(c.mu);
      --c.users;
      if (c.last == ev_spin) c.last = nullptr;
    }
    if (ev_spin) (void)hipEventDestroy(ev_spin);
    ev_spin = nullptr;
    spin_user = false;
  }
  // around the spinning launches of one call (held while they are enqueued: the chain is a total order)
  struct SpinLink {
    ScsHipWork *w;
    SpinChain &c;
    std::unique_lock<std::mutex> lk;
    explicit SpinLink(ScsHipWork *w_) : w(w_), c(spin_chain(w_->device)) {
      w->spin_register();
      lk = std::unique_lock<std::mutex>(c.mu);
      if (c.users >= 2 && c.last && c.last_stream != w->stream) HIP_CHECK(hipStreamWaitEvent(w->stream, c.last, 0));
    }
    ~SpinLink() {
      if (c.users >= 2 && hipEventRecord(w->ev_spin, w->stream) == hipSuccess) {
        c.last = w->ev_spin;
        c.last_stream = w->stream;
      }
    }
  };
  int psd_mc_members(int big) {
    if (psd_mc_cap < 0) {
      int coop = 0, per_cu = 0, cus = 0;
      (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_psd_sweep_mc), kPsdThreads, kPsdMcLdsBytes) != hipSuccess)
        per_cu = 0;
      psd_mc_cap = coop ? std::min(per_cu, 1) * cus : 0;  // one member per CU: the pivot solves want a SIMD each
    }
    const int groups = 8 * ((big + 7) / 8);
    // at least ~3 pivots per member and step: below that the barriers cost more than the spread saves
    // (tools/psd_mc_lab.sh: order 200 x 50, G = 4: 3.57 -> 2.24 ms per projection; order 64 x 100, G = 2: 0.29 -> 0.42 ms)
    const int pivots = psd_max_np / (2 * kPsdB);
    int G = std::min(std::min(psd_mc_cap / groups, kPsdMcMaxG), pivots / 3);
    if (psd_mc_forced >= 0) {
      G = psd_mc_forced;
      if (G > kPsdMcMaxG || ((long)G * groups > (long)psd_mc_cap && !psd_mc_nocheck)) G = 1;  // (NOCHECK: tests of the refused launch)
    }
    return std::max(G, 1);
  }
  bool in_capture = false;
  // stopping level of the PSD sweeps (psd.hpp psd_offtol2): inside the ADMM loop the iteration's P_PSD_TOL2, else nullptr = fixed 1e-8
  const double *psd_tol2 = nullptr;
  static bool psd_tol_adaptive() { return opts().psd_tol_adaptive; }  // SCS_HIP_PSD_TOL=fixed: A/B
  // ... and only while no Anderson extrapolation can happen yet (the history is still filling: iteration < lookback x interval;
  // always, without acceleration): plain ADMM tolerates inexact projections, the secant model of the acceleration does not —
  // with interval 1 and type-II steps a golden infeasible instance stalled for good (tools/dbg/psd_tol_infeas.py).
  double psd_tol2_for(int iter) const {
    const bool plain_phase = aa.mem <= 0 || (long)iter < (long)aa.mem * stgs.acceleration_interval;
    return plain_phase ? psd_tol2_of(psd_res_min) : kPsdOffTol2;
  }
  static double psd_kappa() {
    return opts().psd_tol_k;  // (labs knob; see psd.hpp psd_offtol2 for why 1e-2)
  }
  static double psd_tol2_of(double level) {  // level = what note_check_residuals left in psd_res_min
    if (!psd_tol_adaptive()) return kPsdOffTol2;
    const double cap = opts().psd_tol_max;  // (labs knob)
    const double t = std::min(std::max(level, 1e-8), cap);
    return t * t;
  }
  int psd_warm = 1;  // warm-start the eigen-solves from the previous call's eigenvectors (0 in the one-shot test entry)

  // AA (aa.hpp): f = v (map output), x = v_prev (map input); the safeguard verdict rides along with the CG flags
  DeviceAa aa;
  double aa_norm = 0;
  int rejected_accel = 0, accepted_accel = 0;

  // per-solve state
  Residuals r;
  double sum_log_scale_factor = 0;
  int n_log_scale_factor = 0, last_scale_update_iter = 0, scale_updates = 0;
  long tot_cg_iters = 0;
  int last_cg_iters = 8;
  int cg_hist[8] = {8, 8, 8, 8, 8, 8, 8, 8}, cg_hist_pos = 0;  // CG steps of the last 8 linear solves (chunk sizing)
  void note_cg_iters(int it) { cg_hist[cg_hist_pos++ & 7] = it; }
  // largest step count of the last `chunk_window()` linear solves (SCS_HIP_CHUNK_WINDOW, 1..8): what a queued iteration's CG chunk is sized
  // for.  Round 4: 3 instead of 8 — in the cold-start phase the counts FALL from iteration to iteration, and a window of 8 kept
  // enqueuing the counts of eight iterations ago: 36 % of the K1 / K2 launches of the bench window were early-exit launches
  // (profiles/r03_bench_kernel_trace.txt: 3581 launched, 2309 with work).
  static int chunk_window() {
    return opts().chunk_window;  // (labs knob)
  }
  int recent_cg_max() const {
    int mx = 1;
    for (int k = 1; k <= chunk_window(); ++k) mx = std::max(mx, cg_hist[(cg_hist_pos - k) & 7]);
    return mx;
  }
  int recent_cg_q3() const {  // third quartile of the last 8 linear solves (the grouped loop's prediction: a short round is cheap there)
    int h[8];
    std::copy(cg_hist, cg_hist + 8, h);
    std::sort(h, h + 8);
    return std::max(1, h[5]);
  }
  double cg_res_min = 0;
  // what the PSD stopping level follows (psd_tol2_of): the smallest of the residuals ANY termination test looks at —
  // primal / dual residual and, for a problem drifting towards a certificate, the certificate's own residuals
  double psd_res_min = 0;
  void note_check_residuals() {
    cg_res_min = std::min(r.nm_pri_n, r.nm_dual_n);
    psd_res_min = psd_kappa() * cg_res_min;
    if (std::isfinite(r.res_infeas)) psd_res_min = std::min(psd_res_min, r.res_infeas);
    // (an unboundedness certificate needs BOTH of its residuals small; |Px| / -c'x is identically 0 for an LP)
    if (std::isfinite(r.res_unbdd_a) && std::isfinite(r.res_unbdd_p)) psd_res_min = std::min(psd_res_min, std::max(r.res_unbdd_a, r.res_unbdd_p));
  }
  // live kernel timing (HIP events on the launch stream, one sampled CG step per chunk)
  bool profile = false;
  double prof_ms[2] = {0, 0};  // K1 (A p), K2 (A' z [+P])
  long prof_n[2] = {0, 0};
  // bench.py: a timestamp INSIDE a solve (scs_hip_set_mark): when iteration mark_iter is about to start the stream is
  // drained and the elapsed time / counters are recorded, so a window that starts past the cold start can be timed
  int mark_iter = -1;
  double mark_ms = -1;
  long mark_cg = 0;
  int mark_aa_calls = 0, mark_aa_accept = 0;
  std::mutex mtx;

  ~ScsHipWork() {
    // nothing of this workspace is in flight once its stream is idle: its blocks may be handed to the next workspace without hipFree
    // (real workspaces only: the stack workspaces of the kernel-level entry points borrow a stream that is gone by now)
    if (stream && (pooled_stream || owns_stream) && hipStreamSynchronize(stream) == hipSuccess) {
      pool_window_end.armed = true;
      ++t_pool_release;
    }
    spin_unregister();
    for (auto &g : g_pre) if (g) (void)hipGraphExecDestroy(g);
    for (auto &g : g_cg) if (g) (void)hipGraphExecDestroy(g);
    if (g_post) (void)hipGraphExecDestroy(g_post);
    if (pinned_block) {
      g_pinned.release(pinned_block);
    } else {  // (stack workspaces of the kernel-level entry points allocate what they need themselves)
      if (h_pin) (void)hipHostFree(h_pin);
      if (h_flags) (void)hipHostFree(h_flags);
      if (h_params_base) (void)hipHostFree(h_params_base);
      for (auto &hf : h_flags_slot) if (hf) (void)hipHostFree(hf);
    }
    if (sol_pin) (void)hipHostFree(sol_pin);
    for (auto &e : sol_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : ev_iter) if (e) (void)hipEventDestroy(e);
    for (auto &es : ev_prof) for (auto &e : es) if (e) (void)hipEventDestroy(e);
    for (auto &es : ev_cone) for (auto &e : es) if (e) (void)hipEventDestroy(e);
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    if (stream && pooled_stream) g_streams.release(device, stream);
    else if (stream && owns_stream) (void)hipStreamDestroy(stream);
  }

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
      if (has_P) launch_spmv(Pf.view(), cg_p.p, EpiStore{cg_Gp.p, 0}, fl.p + F_DONE, stream);
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
        if (profile && h_flags[F_ITERS] - iters_before > sample_it) {  // the sampled step really ran
          float a = 0, b = 0;
          if (hipEventElapsedTime(&a, ev[0], ev[1]) == hipSuccess && hipEventElapsedTime(&b, ev[1], ev[2]) == hipSuccess) {
            prof_ms[0] += a; prof_n[0]++;
            prof_ms[1] += b; prof_n[1]++;
          }
        }
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
    if (profile && prof_step[slot] >= 0 && last_cg_iters > prof_step[slot]) {  // the sampled step really ran
      float a = 0, b = 0;
      if (hipEventElapsedTime(&a, ev_prof[slot][0], ev_prof[slot][1]) == hipSuccess &&
          hipEventElapsedTime(&b, ev_prof[slot][1], ev_prof[slot][2]) == hipSuccess) {
        prof_ms[0] += a; prof_n[0]++;
        prof_ms[1] += b; prof_n[1]++;
      }
    }
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
    prof_ms[0] = prof_ms[1] = 0;
    prof_n[0] = prof_n[1] = 0;
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
};

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

// ---- write_data_filename (kwarg R:scs/scsobject.h:493,550; tests R:test/test_scs_coverage.py:532-537,1728-1738) ----
// Self-describing little-endian dump of (settings, cone, data) taken BEFORE equilibration, so that an instance
// can be replayed.  Layout: magic "SCSHIP01", then records  <u32 tag><u64 count><payload>  with tags
// 1 dims(i32 m,n) 2 settings(f64 x 16, field order of ScsSettings without the file names) 3 cone scalars (i32 z,l,bsize,ep,ed)
// 4 bu 5 bl 6 q 7 s 8 p 9 b 10 c 11 A.x 12 A.i 13 A.p 14 P.x 15 P.i 16 P.p 17 cs  (f64 or i32 arrays).
static void write_record(FILE *f, unsigned tag, const void *ptr, size_t count, size_t elem) {
  const unsigned long long c = count;
  std::fwrite(&tag, sizeof(tag), 1, f);
  std::fwrite(&c, sizeof(c), 1, f);
  if (count) std::fwrite(ptr, elem, count, f);
}
static void write_problem_data(const char *fname, const ScsData *d, const ScsCone *k, const ScsSettings *st) {
  FILE *f = std::fopen(fname, "wb");
  if (!f) return;  // like the reference: a diagnostics file that cannot be opened is not fatal
  std::fwrite("SCSHIP01", 1, 8, f);
  const int dims[2] = {d->m, d->n};
  write_record(f, 1, dims, 2, sizeof(int));
  const double sv[16] = {(double)st->normalize, st->scale, (double)st->adaptive_scale, st->rho_x, (double)st->max_iters,
                         st->eps_abs, st->eps_rel, st->eps_infeas, st->alpha, st->time_limit_secs, (double)st->verbose,
                         (double)st->acceleration_lookback, (double)st->acceleration_interval,
                         (double)st->acceleration_type_1, st->acceleration_regularization, st->acceleration_relaxation};
  write_record(f, 2, sv, 16, sizeof(double));
  const int cs[5] = {k->z, k->l, k->bsize, k->ep, k->ed};
  write_record(f, 3, cs, 5, sizeof(int));
  const size_t nb = k->bsize > 1 ? (size_t)k->bsize - 1 : 0;
  write_record(f, 4, k->bu, nb, sizeof(double));
  write_record(f, 5, k->bl, nb, sizeof(double));
  write_record(f, 6, k->q, (size_t)k->qsize, sizeof(int));
  write_record(f, 7, k->s, (size_t)k->ssize, sizeof(int));
  write_record(f, 8, k->p, (size_t)k->psize, sizeof(double));
  if (k->cssize) write_record(f, 17, k->cs, (size_t)k->cssize, sizeof(int));
  write_record(f, 9, d->b, (size_t)d->m, sizeof(double));
  write_record(f, 10, d->c, (size_t)d->n, sizeof(double));
  write_record(f, 11, d->A->x, (size_t)d->A->p[d->n], sizeof(double));
  write_record(f, 12, d->A->i, (size_t)d->A->p[d->n], sizeof(int));
  write_record(f, 13, d->A->p, (size_t)d->n + 1, sizeof(int));
  if (d->P) {
    write_record(f, 14, d->P->x, (size_t)d->P->p[d->n], sizeof(double));
    write_record(f, 15, d->P->i, (size_t)d->P->p[d->n], sizeof(int));
    write_record(f, 16, d->P->p, (size_t)d->n + 1, sizeof(int));
  }
  std::fclose(f);
}

// ---- log_csv_filename: one row per ADMM iteration, the 36 columns of the reference's logs
// (R:notebooks/analyze_csv_logs.ipynb cell 3; kwarg R:scs/scsobject.h:494,551; tests R:test/test_scs_coverage.py:540-547,1739-1751)
static const char *kCsvHeader =
    "iter,res_pri,res_dual,gap,ax_s_btau_nrm_inf,px_aty_ctau_nrm_inf,ax_s_btau_nrm_2,px_aty_ctau_nrm_2,res_infeas,"
    "res_unbdd_a,res_unbdd_p,pobj,dobj,tau,kap,res_pri_normalized,res_dual_normalized,gap_normalized,"
    "ax_s_btau_nrm_inf_normalized,px_aty_ctau_nrm_inf_normalized,ax_s_btau_nrm_2_normalized,"
    "px_aty_ctau_nrm_2_normalized,res_infeas_normalized,res_unbdd_a_normalized,res_unbdd_p_normalized,"
    "pobj_normalized,dobj_normalized,tau_normalized,kap_normalized,scale,diff_u_ut_nrm_2,diff_v_v_prev_nrm_2,"
    "diff_u_ut_nrm_inf,diff_v_v_prev_nrm_inf,aa_norm,time,\n";

static void write_csv_row(FILE *f, int iter, const Residuals &r, double scale, const double *diffs, double aa_norm,
                          double time_s) {
  const double nan = NAN;
  // normalised-space counterparts (tau is scale-free)
  const double res_pri_n = safediv_pos(r.nm_pri_n, r.tau), res_dual_n = safediv_pos(r.nm_dual_n, r.tau);
  const double bty_n = safediv_pos(r.bty_tau_n, r.tau), ctx_n = safediv_pos(r.ctx_tau_n, r.tau);
  const double xpx_n = safediv_pos(r.xt_p_x_tau_n, r.tau * r.tau);
  const double gap_n = std::fabs(xpx_n + ctx_n + bty_n), pobj_n = xpx_n / 2. + ctx_n, dobj_n = -xpx_n / 2. - bty_n;
  const double infeas_n = r.bty_tau_n < 0 ? safediv_pos(r.nm_aty_n, -r.bty_tau_n) : nan;
  const double unb_a_n = r.ctx_tau_n < 0 ? safediv_pos(r.nm_ax_s_n, -r.ctx_tau_n) : nan;
  const double unb_p_n = r.ctx_tau_n < 0 ? safediv_pos(r.nm_px_n, -r.ctx_tau_n) : nan;
  const double vals[35] = {r.res_pri, r.res_dual, r.gap, r.nm_ax_s_btau, r.nm_px_aty_ctau, std::sqrt(r.sq_pri_o),
                           std::sqrt(r.sq_dual_o), r.res_infeas, r.res_unbdd_a, r.res_unbdd_p, r.pobj, r.dobj, r.tau, r.kap,
                           res_pri_n, res_dual_n, gap_n, r.nm_pri_n, r.nm_dual_n, std::sqrt(r.sq_pri_n), std::sqrt(r.sq_dual_n),
                           infeas_n, unb_a_n, unb_p_n, pobj_n, dobj_n, r.tau, r.kap_n, scale, std::sqrt(diffs[0]),
                           std::sqrt(diffs[1]), diffs[2], diffs[3], aa_norm, time_s};
  std::fprintf(f, "%d,", iter);
  for (double v : vals) std::fprintf(f, "%.16e,", v);
  std::fprintf(f, "\n");
}

// linsys: 0 = what SCS_HIP_LINSYS says (default indirect), 1 = indirect (PCG), 2 = dense direct (dense.hpp)
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
 * pair around each batch (event overhead amortised); out = {K1 avg ms, K2 avg ms, K3 avg ms (0 without P)} */
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
    if (w->has_P) {  // K3: Gp = P p (csrc/spmv*.hpp on the full symmetric CSR of P, epilogue EpiStore)
      HIP_CHECK(hipEventRecord(w->ev[0], s));
      for (int i = 0; i < reps; ++i) launch_spmv(w->Pf.view(), w->cg_p.p, EpiStore{w->cg_Gp.p, 0}, nullptr, s);
      HIP_CHECK(hipEventRecord(w->ev[1], s));
      HIP_CHECK(hipEventSynchronize(w->ev[1]));
      HIP_CHECK(hipEventElapsedTime(&c, w->ev[0], w->ev[1]));
    }
    out[0] = a / reps;
    out[1] = b / reps;
    out[2] = c / reps;
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
}

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

}  // extern "C"
