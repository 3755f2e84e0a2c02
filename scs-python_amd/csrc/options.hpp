// options.hpp — every environment variable the library reads, in ONE place (round 6).
//
// The product has 17 runtime knobs (INTEGRATION.md "Runtime knobs"; the Python loader adds SCS_HIP_LIB, SCS_HIP_RUNTIME and shares
// SCS_HIP_RUNTIME_ENV).  They are parsed HERE, once per scs_init / kernel-level entry point (refresh_options(): a test may change the
// environment between two workspaces of one process; a workspace keeps what it was created with), never at the point of use.
// Everything else that rounds 1-5 could switch at run time — the experiments that lost (K2 without p, MINRES, the persistent CG kernel,
// hipGraph replay, cooperative launches, the in-kernel combine of split layouts, the gather-ahead schedule ...) and the lab switches of
// the kernels — is a compile-time default in the product and only reads the environment in the `-DSCS_HIP_LABS` build
// (scs-python_amd/Makefile `make labs` -> libscs_hip_labs.so, used by tools/ and by the tests marked `labs`).
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

extern char **environ;

namespace scship {

enum : unsigned { DBG_PIPE = 1u, DBG_GROUP = 2u, DBG_SETUP = 4u, DBG_TOL = 8u };

struct Options {
  // ------------------------------------------------------------------ product knobs (per workspace)
  bool pipeline = true;        // SCS_HIP_PIPELINE=0: the host looks at the CG flags in every iteration (no run-ahead)
  int pipe_chunk = 0;          // SCS_HIP_PIPELINE=N (N >= 2): run-ahead with CG chunks of N steps (tests: forces stalls)
  bool host_setup = false;     // SCS_HIP_SETUP=host: transposition / layout builders on the host (fallback, entry-for-entry identical)
  bool lazy_setup = true;      // SCS_HIP_LAZY_SETUP=0: small problems finish R, the preconditioner / G^-1 and g inside scs_init
  bool cs = true;              // SCS_HIP_CS=0: never the column-sorted pass layout (slab / CSR-stream kernels)
  long cs_min_nnz = 1L << 20;  // SCS_HIP_CS=N (N >= 2): the pass layout from N nonzeros on (tests on small matrices)
  bool cs_split = true;        // SCS_HIP_CS_SPLIT=0: one workgroup per row chunk for A' too (bit-exact sequential row sums)
  bool slab = true;            // SCS_HIP_SLAB=0: no L2-blocked slab copy (and no pass layout): the plain CSR-stream kernel
  bool aa_gram = false;        // SCS_HIP_AA=gram: Anderson acceleration through the Gram matrix + host LU instead of the TSQR
  int psd_split = -1;          // SCS_HIP_PSD_SPLIT=0|1: force the one-launch / the split pipeline of the PSD projection
  int psd_mc = -1;             // SCS_HIP_PSD_MC=G: workgroups per matrix of the multi-CU sweep kernel (1: never spin)
  bool psd_refine = true;      // SCS_HIP_PSD_REFINE=0: strict Jacobi sweeps only (bit-identical to the one-launch kernel)
  bool psd_tol_adaptive = true;  // SCS_HIP_PSD_TOL=fixed: sweeps always to 1e-8
  int spin_budget_log2 = 25;   // SCS_HIP_SPIN_BUDGET_LOG2: barrier polls before a spinning kernel gives up (tests: 0)
  bool linsys_dense = false;   // SCS_HIP_LINSYS=dense: the bare C entry scs_init builds the dense direct solver
  unsigned debug = 0;          // SCS_HIP_DEBUG=pipe,group,setup,tol: diagnostics on stderr
  // (process-wide, read once where they are used: SCS_HIP_STREAMS, SCS_HIP_POOL_MB, SCS_HIP_CTRLC, SCS_HIP_RUNTIME_ENV)

  // ------------------------------------------------------------------ labs (compile-time defaults in the product)
  bool k1dot = false;          // p'Gp from K1 (cg_k1dot.hpp): K2 3 us faster, the iteration 1.7 % slower
  int krylov = 0;              // 1 MINRES, 2 auto (minres.hpp): 2.1 x slower on whole config-3 solves
  double mr_tolf = 1.0;
  bool mr_check = false;
  int persist_w = 0, persist_g = 1;  // persistent one-launch CG (cg_persist.hpp): not faster than launch-per-kernel
  bool graph = false;          // hipGraph replay of the iteration (SCS_HIP_PIPELINE=0 only): 5 % slower than eager at config 2
  long graph_max_l = 1000000L;
  bool psd_coop = false;       // hipLaunchCooperativeKernel for the multi-CU sweeps: ~2 ms per launch in a warm process
  int cs_sched = 3;            // 1 gather-ahead kernel, 2 braid, 3 braid + stream loads before the barrier (shipped)
  bool cs_combine = false;     // in-kernel combine of split layouts: eats the gather gain
  int cs_rpt = 0, cs_split_a = 0, cs_split_at = 0, cs_split_p = 0;
  bool cs_peel = true, cs_peel_ladder = true, cs_virt = true;
  int slab_rpt = 0, slab_shift = 17;
  bool arena = true, cg_fuse = true, norm_fuse = true, soc_psd_fuse = true, dense_full_gemv = true, aa_fast = true, wait_spin = false;
  int chunk_window = 3, group_predict = 1, group_max = 1024, group_lanes = 1, group_min = 16, aa_waves1 = 2048;
  double psd_gate_k = 0., psd_gate_off = 0., psd_gate_omega = 0., psd_tol_k = 1e-2, psd_tol_max = 1e-3;
  bool psd_la = true, psd_mc_nocheck = false, psd_small_one_wave = false;
  bool pool_poison = false;

  static bool is0(const char *e) { return e && e[0] == '0'; }
  static bool is1(const char *e) { return e && e[0] == '1'; }
  static Options parse() {
    Options o;
    if (const char *e = getenv("SCS_HIP_PIPELINE")) {
      const int v = atoi(e);
      o.pipeline = !is0(e);
      o.pipe_chunk = v >= 2 ? v : 0;
    }
    if (const char *e = getenv("SCS_HIP_SETUP")) o.host_setup = e[0] == 'h';
    o.lazy_setup = !is0(getenv("SCS_HIP_LAZY_SETUP"));
    if (const char *e = getenv("SCS_HIP_CS")) {
      const long v = atol(e);
      o.cs = !is0(e);
      if (v >= 2) o.cs_min_nnz = v;
    }
    o.cs_split = !is0(getenv("SCS_HIP_CS_SPLIT"));
    o.slab = !is0(getenv("SCS_HIP_SLAB"));
    if (const char *e = getenv("SCS_HIP_AA")) o.aa_gram = e[0] == 'g';
    if (const char *e = getenv("SCS_HIP_PSD_SPLIT")) o.psd_split = e[0] == '1' ? 1 : 0;
    if (const char *e = getenv("SCS_HIP_PSD_MC")) o.psd_mc = atoi(e);
    o.psd_refine = !is0(getenv("SCS_HIP_PSD_REFINE"));
    if (const char *e = getenv("SCS_HIP_PSD_TOL")) o.psd_tol_adaptive = e[0] != 'f';
    if (const char *e = getenv("SCS_HIP_SPIN_BUDGET_LOG2")) { const int v = atoi(e); o.spin_budget_log2 = v < 0 ? 0 : (v > 40 ? 40 : v); }
    if (const char *e = getenv("SCS_HIP_LINSYS")) o.linsys_dense = e[0] == 'd' || e[0] == 'D';
    if (const char *e = getenv("SCS_HIP_DEBUG")) {
      if (strstr(e, "pipe")) o.debug |= DBG_PIPE;
      if (strstr(e, "group")) o.debug |= DBG_GROUP;
      if (strstr(e, "setup")) o.debug |= DBG_SETUP;
      if (strstr(e, "tol")) o.debug |= DBG_TOL;
      if (strstr(e, "all")) o.debug = ~0u;
    }
#ifdef SCS_HIP_LABS
    auto pos_int = [](const char *name, int dflt) { const char *e = getenv(name); const int v = e ? atoi(e) : 0; return v > 0 ? v : dflt; };
    auto pos_dbl = [](const char *name, double dflt) { const char *e = getenv(name); const double v = e ? atof(e) : 0.; return v > 0. ? v : dflt; };
    o.k1dot = is1(getenv("SCS_HIP_K1DOT"));
    if (const char *e = getenv("SCS_HIP_KRYLOV")) o.krylov = e[0] == 'm' ? 1 : e[0] == 'a' ? 2 : 0;
    o.mr_tolf = pos_dbl("SCS_HIP_MR_TOLF", 1.0);
    o.mr_check = getenv("SCS_HIP_MR_CHECK") != nullptr;
    if (const char *e = getenv("SCS_HIP_PERSIST")) {  // "W" or "WxG": W workgroups of G x 256 lanes
      o.persist_w = atoi(e);
      if (const char *x = strchr(e, 'x')) o.persist_g = atoi(x + 1);
    }
    o.graph = !is0(getenv("SCS_HIP_GRAPH"));
    if (const char *e = getenv("SCS_HIP_GRAPH_MAX_L")) o.graph_max_l = atol(e);
    o.psd_coop = is1(getenv("SCS_HIP_PSD_COOP"));
    if (const char *e = getenv("SCS_HIP_CS_SCHED")) o.cs_sched = atoi(e);
    o.cs_combine = is1(getenv("SCS_HIP_CS_COMBINE"));
    o.cs_rpt = pos_int("SCS_HIP_CS_RPT", 0);
    o.cs_split_a = pos_int("SCS_HIP_CS_SPLIT_A", 0);
    o.cs_split_at = pos_int("SCS_HIP_CS_SPLIT_AT", 0);
    o.cs_split_p = pos_int("SCS_HIP_CS_SPLIT_P", 0);
    o.cs_peel = !is0(getenv("SCS_HIP_CS_PEEL"));
    o.cs_peel_ladder = !is0(getenv("SCS_HIP_CS_PEEL_LADDER"));
    o.cs_virt = !is0(getenv("SCS_HIP_CS_VIRT"));
    o.slab_rpt = pos_int("SCS_HIP_SLAB_RPT", 0);
    { const int v = pos_int("SCS_HIP_SLAB_SHIFT", 17); o.slab_shift = (v >= 10 && v <= 24) ? v : 17; }
    o.arena = !is0(getenv("SCS_HIP_ARENA"));
    o.cg_fuse = !is0(getenv("SCS_HIP_CG_FUSE"));
    o.norm_fuse = !is0(getenv("SCS_HIP_NORM_FUSE"));
    o.soc_psd_fuse = !is0(getenv("SCS_HIP_SOC_PSD_FUSE"));
    if (const char *e = getenv("SCS_HIP_DENSE_GEMV")) o.dense_full_gemv = e[0] != 'h';
    o.aa_fast = !is0(getenv("SCS_HIP_AA_FAST"));
    if (const char *e = getenv("SCS_HIP_WAIT")) o.wait_spin = e[0] == 's';
    { const int v = pos_int("SCS_HIP_CHUNK_WINDOW", 3); o.chunk_window = v < 1 ? 1 : (v > 8 ? 8 : v); }
    if (const char *e = getenv("SCS_HIP_GROUP_PREDICT")) o.group_predict = atoi(e);
    o.group_max = pos_int("SCS_HIP_GROUP_MAX", 1024);
    o.group_lanes = pos_int("SCS_HIP_GROUP_LANES", 1);
    o.group_min = pos_int("SCS_HIP_GROUP_MIN", 16);
    o.aa_waves1 = pos_int("SCS_HIP_AA_WAVES1", 2048);
    o.psd_gate_k = pos_dbl("SCS_HIP_PSD_GATE_K", 0.);
    o.psd_gate_off = pos_dbl("SCS_HIP_PSD_GATE_OFF", 0.);
    o.psd_gate_omega = pos_dbl("SCS_HIP_PSD_GATE_OMEGA", 0.);
    o.psd_tol_k = pos_dbl("SCS_HIP_PSD_TOL_K", 1e-2);
    o.psd_tol_max = pos_dbl("SCS_HIP_PSD_TOL_MAX", 1e-3);
    o.psd_la = !is0(getenv("SCS_HIP_PSD_LA"));
    o.psd_mc_nocheck = getenv("SCS_HIP_PSD_MC_NOCHECK") != nullptr;
    o.psd_small_one_wave = is1(getenv("SCS_HIP_PSD_SMALL_WAVES"));
    o.pool_poison = is1(getenv("SCS_HIP_POOL_POISON"));
#endif
    return o;
  }
};

// The current options live behind an atomic pointer: readers (opts(), anywhere, any thread) never see a half-written struct, and a
// reference they hold stays valid — a superseded struct is never freed (a few hundred bytes per CHANGE of the environment, which only
// tests make).
inline std::atomic<const Options *> &options_ptr() {
  static std::atomic<const Options *> p{new Options(Options::parse())};
  return p;
}
// what the NEXT workspace / kernel-level call sees (the environment as it was at the last refresh_options())
inline const Options &opts() { return *options_ptr().load(std::memory_order_acquire); }
// re-read the environment: first thing in scs_init and in every kernel-level entry point.  A new struct is only published when an
// SCS_HIP_* variable changed since the last call, so concurrent scs_init calls under one environment publish nothing.
inline void refresh_options() {
  static std::mutex m;
  static std::string last = "\x01";  // (never equal to a real signature: the first call publishes what it parsed)
  std::string sig;
  for (char **e = environ; e && *e; ++e)
    if (std::strncmp(*e, "SCS_HIP_", 8) == 0) { sig += *e; sig += '\n'; }
  std::lock_guard<std::mutex> lk(m);
  if (sig != last) {
    options_ptr().store(new Options(Options::parse()), std::memory_order_release);
    last = sig;
  }
}
#ifdef SCS_HIP_LABS
constexpr bool kLabsBuild = true;
#else
constexpr bool kLabsBuild = false;
#endif

}  // namespace scship
