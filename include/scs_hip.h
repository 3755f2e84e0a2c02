/*
 * scs_hip.h — C-ABI of libscs_hip.so, the MI355X-native (gfx950) implementation
 * of SCS's ADMM hot path.  Plain pointers and sizes only; no torch types.
 *
 * PART 1 is exactly the core API the reference's CPython glue binds
 * (SURVEY.md §8 row b6); a maintainer can re-compile R:scs/scsobject.h against
 * this header + scs_types.h to obtain `scs._scs_hip`.
 * PART 2 are kernel-level entry points used by the parity tests and bench.py
 * (they have no counterpart in the reference's public API; each cites the
 * absent upstream file whose role it plays).
 */
#ifndef SCS_HIP_H_GUARD
#define SCS_HIP_H_GUARD

#include "scs_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ScsHipWork ScsWork;

/* ------------------------------------------------------------------ PART 1 */

/* replaces scs_init — called at R:scs/scsobject.h:903.  Copies all of d,k,stgs
 * (the glue frees its views right after, :908).  Builds CSR(A) next to the
 * caller's CSC(A), equilibrates, uploads everything to HBM, pre-solves g.
 * Returns NULL on invalid data / allocation failure / no usable GPU. */
ScsWork *scs_init(const ScsData *d, const ScsCone *k, const ScsSettings *stgs);

/* replaces scs_solve — called at R:scs/scsobject.h:986 with the GIL released.
 * Runs the whole ADMM loop device-resident; only info scalars and the final
 * x,y,s cross PCIe.  sol holds the warm start on entry when warm_start != 0. */
scs_int scs_solve(ScsWork *w, ScsSolution *sol, ScsInfo *info, scs_int warm_start);

/* replaces scs_update — called at R:scs/scsobject.h:1217.  b and/or c may be NULL. */
scs_int scs_update(ScsWork *w, scs_float *b, scs_float *c);

/* replaces scs_finish — called at R:scs/scsobject.h:1240. */
void scs_finish(ScsWork *w);

/* replaces scs_set_default_settings — called at R:scs/scsobject.h:520. */
void scs_set_default_settings(ScsSettings *stgs);

/* replaces scs_version — called at R:scs/scsmodule.h:5. */
const char *scs_version(void);

/* sizeof(scs_int), sizeof(scs_float): what R:scs/scsmodule.h:16-23 report. */
size_t scs_sizeof_int(void);
size_t scs_sizeof_float(void);

/* ------------------------------------------------------------------ PART 2 */

/* number of visible HIP devices (0 => library unusable); does not initialise a context */
int scs_hip_device_count(void);
/* choose the device used by subsequent scs_init calls (and the kernel-level entry points below): the process-wide
 * default (0 at start), or — set_thread_device, dev < 0 clears it — a default of the calling thread only.  A workspace
 * remembers the device it was created on: scs_solve / scs_update / scs_finish select it themselves, so one process
 * may drive several GPUs. */
int scs_hip_set_device(int dev);
int scs_hip_set_thread_device(int dev);
/* 1 when this library is the -DSCS_HIP_LABS build (scs-python_amd/Makefile `make labs`: the experiments that lost their measurement and
 * the lab switches of the kernels compiled in and readable from the environment — csrc/options.hpp), 0 for the product. */
int scs_hip_labs_build(void);
/* free and total HBM bytes of the device subsequent scs_init calls would use, plus the bytes this library's block pool holds for
 * reuse (they count as free for a new workspace).  What the Python layer's `LinearSolver.AUTO` asks before it picks the dense direct
 * solver (R:scs/py/__init__.py:45-54 resolves AUTO to the best DIRECT backend that is usable).  0 on success, -1 without a device. */
int scs_hip_mem_info(size_t *free_bytes, size_t *total_bytes);

/* y (+)= A x or A' x through the hot-path SpMV kernels (row a3; plays the role
 * of scs_source/linsys/scs_matrix.c accum_by_a / accum_by_atrans, R:meson.build:199-202).
 * A is CSC with int32 indices; x,y are host pointers.  Returns 0 on success. */
int scs_hip_spmv(const ScsMatrix *A, const scs_float *x, scs_float *y, int transpose);

/* HOST-ONLY check of the column-sorted pass layout the large-matrix SpMV kernels read (spmv_cs.hpp): builds the
 * layout of A (transpose=0) or A' (transpose=1) with the host builder (rows per lane `rpt` = 1, 2, 4, 8, 16, or 0 =
 * the geometry scs_init would pick; `split` = 1, or 2 workgroups per row chunk as scs_init uses for A') and
 * evaluates y += M x by walking it exactly as the kernel does (slot scatter, per-lane runs, pass order, partial sums).
 * No GPU needed.  Returns 0, 1 if the pattern does not fit the format, -1 on error. */
int scs_hip_cs_layout_host_spmv(const ScsMatrix *A, const scs_float *x, scs_float *y, int transpose, int rpt, int split);
/* Same walk for the virtual-row variant of that layout (round 3): rows longer than max(piece_len, what a count field
 * holds) are cut into pieces of at most piece_len nonzeros that ride in the passes; the pieces of a row are then added
 * the way the device does it (64 lanes striding over them, shuffle tree).  Returns 1 when no row is that long. */
int scs_hip_cs_layout_host_spmv_pieces(const ScsMatrix *A, const scs_float *x, scs_float *y, int transpose, int piece_len);

/* Time `reps` launches of the A (transpose=0) or A' (transpose=1) SpMV kernel
 * with HIP events on the launch stream; inputs already resident in HBM.
 * Returns average milliseconds per launch, <0 on error.  (bench.py roofline leg) */
double scs_hip_spmv_bench(const ScsMatrix *A, int transpose, int reps);

/* In-place projection of x (length m, host pointer) onto K (dual=0) or K*
 * (dual=1) with the hot-path cone kernels (row a5; scs_source/src/cones.c,
 * exp_cone.c, R:meson.build:188,190). */
int scs_hip_proj_cone(scs_float *x, const ScsCone *k, scs_int m, int dual);

/* The same projection applied to `count` vectors one after the other (xs: count x m, row-major, in place) through ONE set of cone
 * workspaces, warm-started from call to call as inside the ADMM loop (K9: eigenvectors of the previous call, periodic
 * re-orthogonalisation, refinement stage; box cone: the previous t).  stats (may be NULL): scs_hip_psd_refine_stats records of the
 * first stats_cap large PSD matrices after the last call.  Returns the number of records written, -1 on error.  (parity tests) */
int scs_hip_proj_cone_seq(scs_float *xs, const ScsCone *k, scs_int m, int dual, int count, scs_float *stats, int stats_cap);

/* One indirect KKT solve [[R_x+P, A'],[A,-R_y]] z = rhs (in place, length n+m)
 * with the device PCG (row a4; scs_source/linsys/cpu/indirect/private.c,
 * R:meson.build:261).  cg_iters may be NULL. */
int scs_hip_kkt_solve(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, scs_float *rhs,
                      scs_float tol, scs_int *cg_iters);

/* Same system solved with the DENSE DIRECT linsys (csrc/dense.hpp): G = R_x + P + A' R_y^{-1} A is formed and inverted on
 * the device (blocked Gauss-Jordan on the fp64 MFMA), x = G^{-1}(rhs_x + A' R_y^{-1} rhs_y), y = R_y^{-1}(A x - rhs_y).  n <= 8192.
 * Plays the role of the reference's direct backends (QDLDL / cuDSS / the LAPACK dense module: R:meson.build:241-262,374-391). */
int scs_hip_kkt_solve_dense(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r, scs_float *rhs);

/* scs_init with an explicit choice of the linear-system solver: 1 = sparse indirect (PCG; what scs_init builds unless the
 * environment says SCS_HIP_LINSYS=dense), 2 = dense direct (n <= 8192: the explicit inverse of the reduced KKT matrix lives in
 * HBM, the linear solve of an ADMM iteration is three dependent launches and never waits for the host), 0 = the default.
 * The reference selects its linear solver by MODULE (R:scs/py/__init__.py:40-66: _scs_direct, _scs_indirect, _scs_gpu, ...);
 * scs._scs_hip binds 1, scs._scs_hip_dense binds 2.  Everything else is scs_init's contract. */
ScsWork *scs_hip_init_linsys(const ScsData *d, const ScsCone *k, const ScsSettings *stgs, int linsys);
/* 1 / 2 as above for a live workspace (0: NULL) */
int scs_hip_linsys_kind(const ScsWork *w);

/* Equilibrate (A,P,b,c) exactly as scs_init does (row a7; scs_source/src/normalize.c,
 * R:meson.build:192).  A->x, P->x, b, c are overwritten; D (m), E (n), sigma (1) filled. */
int scs_hip_normalize(ScsMatrix *A, ScsMatrix *P, scs_float *b, scs_float *c, const ScsCone *k,
                      scs_float *D, scs_float *E, scs_float *sigma);

/* Measured device-copy bandwidth ceiling in GB/s (float4 copy of `bytes` bytes). */
double scs_hip_copy_bandwidth(size_t bytes, int reps);

/* Live timing of the two dominant kernels inside scs_solve: when enabled, one CG step per
 * host sync is bracketed by HIP events on the solver's own stream.  out[12] =
 * {K1 total ms, K1 samples, K2 total ms, K2 samples, nnz(A), K1 workgroups, K2 workgroups, nnz(P full),
 *  nonlinear cone projections total ms, samples (one per queued iteration), K3 total ms, K3 samples}
 * where K1 = z <- R_y^{-1} A p,  K2 = Gp <- A' z + R_x p (+ P p)  and, for problems with P, K3 = P p (between K1 and K2). */
void scs_hip_set_profiling(ScsWork *w, int on);
void scs_hip_kernel_times(const ScsWork *w, double *out);
/* The final (x, y, s) of the last scs_solve, copied from the workspace's HBM buffers to caller-provided DEVICE pointers
 * (any may be NULL; n, m, m doubles) on the workspace's stream, complete on return.  Same values, bit for bit, as the
 * host copies scs_solve returned (NaN where the status leaves a vector undefined).  scs/batch.py gathers from these. */
int scs_hip_solution_to_device(ScsWork *w, scs_float *x_dev, scs_float *y_dev, scs_float *s_dev);

/* Grouped solve of `count` independent, already initialised workspaces (BASELINE.json configs[4]: a batch of small cone
 * programs; the reference's notion is "independent instances run concurrently", R:test/test_thread_safety.py:78-93 —
 * one scs_solve per thread).  Members of equal shape (n, m, cone structure, Anderson schedule) whose matrices use the
 * CSR-stream layout advance through the ADMM loop in lock step and SHARE every kernel launch (blockIdx.y = problem,
 * arguments from a per-problem record in HBM; csrc/batch.hpp); whatever cannot be grouped is solved by scs_solve's own
 * loop, one after the other.  sol[i] / info[i] are filled exactly as scs_solve(w[i], sol[i], info[i], warm_start) fills
 * them — the grouped kernels run the same device code over the same block decomposition, so iterates, iteration and
 * CG-step counts are bit-identical to separate solves (timing fields are those of the group).  Returns 0, -1 on error
 * (scs_hip_last_error).  The workspaces must live on one device and must not be used by other threads meanwhile. */
scs_int scs_hip_solve_batch(ScsWork **w, ScsSolution **sol, ScsInfo **info, scs_int count, scs_int warm_start);

/* bench.py: a timestamp inside the next scs_solve calls.  When ADMM iteration `iter` is about to start, the stream is
 * drained and out[4] = {ms since the start of the solve, CG steps so far, Anderson calls so far, accepted so far} is
 * recorded (out[0] < 0: the solve ended before that iteration); iter < 0 switches it off. */
void scs_hip_set_mark(ScsWork *w, int iter);
void scs_hip_get_mark(const ScsWork *w, double *out);
/* `reps` back-to-back launches of K1, then of K2 and — problems with P — of K3 on the solver's own stream and HBM-resident
 * data, one HIP event pair per batch (the ~10-20 us per-event overhead is amortised).
 * out[4] = {K1 avg ms, K2 avg ms, K3 avg ms as the CG step runs it — between K1 and K2: (K1, K3, K2) x reps minus (K1, K2) x reps —, K3 avg ms
 * back to back (its matrix may then stay in the Infinity Cache)}; K3: 0 without P.  Returns 0 on success. */
int scs_hip_time_matvec(ScsWork *w, int reps, double *out);

/* Anderson acceleration as a standalone object on host vectors (row a6): the interface of scs_source/src/aa.c
 * (aa_init / aa_apply / aa_safeguard / aa_reset / aa_finish; named at R:meson.build:187, knobs R:README.md:98-104,
 * statistics R:scs/scsobject.h:1096-1107).  mem <= 32.  scs_solve runs the same device object on the resident iterate.
 *   apply:     f = F(x) on entry; on return f may hold the extrapolated iterate.  Returns aa_norm (0: history
 *              still filling, < 0: step rejected and history reset, NaN: error).
 *   safeguard: f_new = F(x_new) of the step after an accepted extrapolation; returns -1 and restores the
 *              pre-extrapolation pair when the fixed-point residual grew, else 0.
 *   last_gamma: weights of the most recent solve (returns their count; gamma may be NULL). */
typedef struct ScsHipAa ScsHipAa;
ScsHipAa *scs_hip_aa_init(scs_int dim, scs_int mem, scs_int type1, scs_float regularization, scs_float relaxation,
                          scs_float safeguard_factor, scs_float max_weight_norm);
scs_float scs_hip_aa_apply(ScsHipAa *a, scs_float *f, const scs_float *x);
scs_int scs_hip_aa_safeguard(ScsHipAa *a, scs_float *f_new, scs_float *x_new);
void scs_hip_aa_reset(ScsHipAa *a);
void scs_hip_aa_get_stats(const ScsHipAa *a, ScsAaStats *st);
scs_int scs_hip_aa_last_gamma(const ScsHipAa *a, scs_float *gamma);
void scs_hip_aa_finish(ScsHipAa *a);

/* bench.py --workload config4_psd: average duration of one batched PSD projection (all `s` cones, warm-started as inside
 * the ADMM loop) on the solver's stream; out[4] = {ms per projection, matrices, largest order, reference flop count
 * ((16/3 + 2) n^3 per matrix)}.  0 on success, 1 when the problem has no PSD cone. */
int scs_hip_time_psd(ScsWork *w, int reps, double *out);

/* Hands the device blocks this library caches for reuse (dead workspaces' buffers, at most SCS_HIP_POOL_MB = 1024 MiB by default) back
 * to the driver: for a process that shares its GPUs with allocators this library does not see (torch, RCCL, other processes). */
void scs_hip_trim_pool(void);

/* How many solves of this process were restarted because a spinning multi-workgroup kernel (multi-CU PSD sweeps, persistent CG)
 * timed out at a barrier — another process held part of the GPU — and were then finished without such kernels (tests). */
long scs_hip_spin_fallbacks(void);

/* K9's refinement stage (csrc/psd.hpp psd_stop_test), diagnostics for tests and bench: for each of the first `cap` PSD matrices of
 * order > 32 of the workspace EIGHT doubles {calls that took the refinement stage so far, refinements whose a-posteriori test sent the
 * matrix back to the sweeps, |K1|_F^2 at the last gate, mixed-sign off-norm^2 / |A|_F^2 after the last refinement, stage flag of the
 * last call (0 none, 1 refined, 2 refined + sweeps), and the gate's view of the last call's matrix as it arrived: |K1|_F^2,
 * |off|_F^2 / |A|_F^2, omega}.  Returns the number of matrices written, -1 on error. */
int scs_hip_psd_refine_stats(ScsWork *w, double *out, int cap);

/* last error message of the calling thread ("" if none) */
const char *scs_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
