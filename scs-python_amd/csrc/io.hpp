// io.hpp — the on-disk formats around the path: write_data_filename dump and the log_csv_filename row
// (one of the units csrc/scs_hip.hip is assembled from — ONE translation unit, in this order: runtime.hpp, device_csr.hpp, work.hpp
// [+ work_linsys.inl, work_admm.inl, work_residuals.inl, work_solve_ends.inl], io.hpp, setup.hpp, loop.hpp, batch.hpp, the C ABI in scs_hip.hip,
// lab_entries.hpp; split out of the 3 800-line file of rounds 1-5 in round 6 — VERDICT r05 item 6 — without moving a line of code)
#pragma once
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
