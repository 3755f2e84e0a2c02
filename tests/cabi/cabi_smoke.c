/* Plain-C consumer of the C-ABI (include/scs_hip.h), the way the reference's CPython glue is one
 * (R:scs/scsobject.h:520,903,986,1217,1240).  min -x  s.t. 0 <= x <= 1 (LP, x* = 1), then the same data
 * over one SOC of dimension 2 (x* = 0.5), then an update of b and a warm-started re-solve; three such LPs through the grouped
 * entry point scs_hip_solve_batch, each compared with its own scs_solve.
 * Build: gcc -O2 -I include tests/cabi/cabi_smoke.c -L scs-python_amd/scs -lscs_hip -Wl,-rpath,... -lm
 * Exit code 0 on success; prints one line per check. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scs_hip.h"

static int check(const char *what, double got, double want, double tol) {
  const int ok = fabs(got - want) <= tol;
  printf("%s: got %.6f want %.6f -> %s\n", what, got, want, ok ? "ok" : "FAIL");
  return ok ? 0 : 1;
}

int main(void) {
  int fails = 0;
  if (scs_hip_device_count() < 1) {
    printf("no HIP device\n");
    return 2;
  }
  printf("scs_version %s sizeof(scs_int)=%zu sizeof(scs_float)=%zu\n", scs_version(), scs_sizeof_int(), scs_sizeof_float());
  scs_float Ax[2] = {1.0, -1.0};
  scs_int Ai[2] = {0, 1}, Ap[2] = {0, 2};
  scs_float b[2] = {1.0, 0.0}, c[1] = {-1.0};
  ScsMatrix A = {Ax, Ai, Ap, 2, 1};
  ScsData d = {2, 1, &A, NULL, b, c};
  ScsSettings st;
  scs_set_default_settings(&st);
  st.verbose = 0;
  st.eps_abs = st.eps_rel = 1e-7;
  scs_float x[1] = {0}, y[2] = {0, 0}, s[2] = {0, 0};
  ScsSolution sol = {x, y, s};
  ScsInfo info;

  ScsCone k;
  memset(&k, 0, sizeof(k));
  k.l = 2;
  ScsWork *w = scs_init(&d, &k, &st);
  if (!w) { printf("scs_init failed: %s\n", scs_hip_last_error()); return 3; }
  scs_int rc = scs_solve(w, &sol, &info, 0);
  fails += (rc != SCS_SOLVED) || strcmp(info.status, "solved") != 0;
  fails += check("LP  x*", x[0], 1.0, 1e-5);
  fails += check("LP  pobj", info.pobj, -1.0, 1e-5);
  /* update b -> 0 <= x <= 2, warm start from the previous solution */
  scs_float b2[2] = {2.0, 0.0};
  fails += scs_update(w, b2, NULL) != 0;
  rc = scs_solve(w, &sol, &info, 1);
  fails += rc != SCS_SOLVED;
  fails += check("LP  x* after update(b)", x[0], 2.0, 1e-5);
  scs_finish(w);

  scs_int q[1] = {2};
  memset(&k, 0, sizeof(k));
  k.q = q;
  k.qsize = 1;
  w = scs_init(&d, &k, &st);
  if (!w) { printf("scs_init failed: %s\n", scs_hip_last_error()); return 3; }
  rc = scs_solve(w, &sol, &info, 0);
  fails += rc != SCS_SOLVED;
  fails += check("SOC x*", x[0], 0.5, 1e-5);
  scs_finish(w);

  /* the grouped entry point (include/scs_hip.h scs_hip_solve_batch; reference notion: independent instances solved
   * concurrently, R:test/test_thread_safety.py:78-93): three workspaces of the LP with different b, solved by ONE call —
   * every sol[i] / info[i] must be what scs_solve(w[i], ...) gives, bit for bit */
  {
    enum { NW = 3 };
    memset(&k, 0, sizeof(k));
    k.l = 2;
    ScsWork *ws[NW];
    scs_float bb[NW][2] = {{1.0, 0.0}, {2.0, 0.0}, {3.5, 0.0}};
    scs_float xs[NW][1], ys[NW][2], ss[NW][2], x1[1], y1[2], s1[2];
    ScsSolution sols[NW], *solp[NW];
    ScsInfo infos[NW], *infop[NW], i1;
    for (int i = 0; i < NW; ++i) {
      ScsData di = {2, 1, &A, NULL, bb[i], c};
      ws[i] = scs_init(&di, &k, &st);
      if (!ws[i]) { printf("scs_init failed: %s\n", scs_hip_last_error()); return 3; }
      sols[i].x = xs[i]; sols[i].y = ys[i]; sols[i].s = ss[i];
      solp[i] = &sols[i]; infop[i] = &infos[i];
    }
    fails += scs_hip_solve_batch(ws, solp, infop, NW, 0) != 0;
    for (int i = 0; i < NW; ++i) {
      char what[64];
      snprintf(what, sizeof(what), "batch member %d x*", i);
      fails += check(what, xs[i][0], bb[i][0], 1e-5);
      ScsSolution so = {x1, y1, s1};
      fails += scs_solve(ws[i], &so, &i1, 0) != SCS_SOLVED;
      const int same = x1[0] == xs[i][0] && y1[0] == ys[i][0] && y1[1] == ys[i][1] && s1[0] == ss[i][0] && s1[1] == ss[i][1] &&
                       i1.iter == infos[i].iter && i1.pobj == infos[i].pobj && strcmp(i1.status, infos[i].status) == 0;
      printf("batch member %d vs scs_solve: %s (%d iterations)\n", i, same ? "identical" : "DIFFERENT", (int)infos[i].iter);
      fails += !same;
      scs_finish(ws[i]);
    }
    fails += scs_hip_solve_batch(NULL, solp, infop, NW, 0) == 0; /* bad arguments are refused, not dereferenced */
  }

  /* invalid cone (dimension mismatch) must come back as NULL, not a crash (R:test/test_scs_basic.py:113-114) */
  memset(&k, 0, sizeof(k));
  k.q = q;
  k.qsize = 1;
  q[0] = 4;
  w = scs_init(&d, &k, &st);
  fails += (w != NULL);
  printf("bad cone -> %s (%s)\n", w ? "workspace?!" : "NULL", scs_hip_last_error());
  printf("%s\n", fails ? "FAILED" : "ALL OK");
  return fails ? 1 : 0;
}
