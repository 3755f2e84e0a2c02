"""GPU tests written the way the reference's own tests are (they read like R:test/test_scs_basic.py,
R:test/test_scs_coverage.py, R:test/test_scs_object.py, R:test/test_thread_safety.py), driving the
public `scs` package with linear_solver=HIP_INDIRECT (and AUTO, which resolves to it)."""
import threading

import numpy as np
import pytest
from numpy.testing import assert_almost_equal
from scipy import sparse as sp

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scs():
    import scs as _scs
    assert _scs._scs_hip.device_count() > 0
    return _scs


BACKENDS = ("hip_indirect", "auto")


def _one_variable_problem():
    """maximise x subject to x <= 1, x >= 0 — written as  min -x,  [1; -1] x + s = [1; 0],  s in K.
    (the one-variable instance the reference's smoke tests use, R:test/test_scs_basic.py:36-72)"""
    return {"A": sp.csc_matrix(np.array([[1.0], [-1.0]])), "b": np.array([1.0, 0.0]), "c": np.array([-1.0])}


data = _one_variable_problem()


@pytest.mark.parametrize("backend", BACKENDS)
def test_one_variable_lp_and_soc(scs, backend):
    """K = R^2_+ : x* = 1.  K = Q^2 (s_1 >= |s_2|, i.e. 1 - x >= x): x* = 1/2."""
    for K, x_star in (({"l": 2, "q": []}, 1.0), ({"l": 0, "q": [2]}, 0.5)):
        out = scs.SCS(data, cone=K, linear_solver=backend, verbose=False).solve()
        assert out["info"]["status"] == "solved"
        assert abs(out["x"][0] - x_star) < 5e-3, (K, out["x"])


def test_bad_calls_are_rejected(scs):
    """argument errors surface as the exception types the reference raises (R:test/test_scs_basic.py:95-114)"""
    for exc, call in (
        (TypeError, lambda: scs.solve()),                                           # no data at all
        (ValueError, lambda: scs.solve(data, {"l": -2, "q": [4]})),                 # negative cone size
        (TypeError, lambda: scs.solve(data, {"l": 2, "q": []}, max_iters=1.1)),     # float where an int is required
        (ValueError, lambda: scs.solve(data, {"l": 0, "q": [1]}, verbose=False)),   # cone dimensions do not add up to m
    ):
        with pytest.raises(exc):
            call()


def test_legacy_solve_with_warm_start_in_data(scs):
    sol = scs.solve(dict(data), {"l": 2}, verbose=False)
    d2 = dict(data, x=sol["x"], y=sol["y"], s=sol["s"])
    sol2 = scs.solve(d2, {"l": 2}, verbose=False)
    assert sol2["info"]["status"] == "solved" and sol2["info"]["iter"] <= sol["info"]["iter"]


def test_qp_known_answer(scs):
    # min 0.5 x^2 - x  st  0 <= x <= 0.5  -> x* = 0.5        (R:test/test_scs_coverage.py:761-779)
    P = sp.csc_matrix([[1.0]])
    d = {"P": P, "A": sp.csc_matrix([[1.0], [-1.0]]), "b": np.array([0.5, 0.0]), "c": np.array([-1.0])}
    sol = scs.SCS(d, {"l": 2}, verbose=False).solve()
    assert sol["info"]["status"] == "solved"
    assert_almost_equal(sol["x"][0], 0.5, decimal=3)


def test_full_P_is_reduced_to_upper_triangle(scs):
    rng = np.random.RandomState(0)
    n, m = 8, 12
    B = rng.randn(n, n)
    P = sp.csc_matrix(B @ B.T + np.eye(n))              # full symmetric
    Am = sp.csc_matrix(rng.randn(m, n))
    d_full = {"P": P, "A": Am, "b": rng.rand(m) + 1.0, "c": rng.randn(n)}
    d_triu = dict(d_full, P=sp.triu(P, format="csc"))
    kw = dict(verbose=False, eps_abs=1e-8, eps_rel=1e-8)
    a = scs.SCS(d_full, {"l": m}, **kw).solve()
    t = scs.SCS(d_triu, {"l": m}, **kw).solve()
    np.testing.assert_array_equal(a["x"], t["x"])


@pytest.mark.parametrize("cone,Adense,bvec,cvec,idx,expected", [
    ({"z": 1}, [[1.0]], [0.7], [1.0], 0, 0.7),                                         # zero cone
    ({"ep": 1}, [[0.0], [0.0], [-1.0]], [1.0, 1.0, 0.0], [1.0], 0, np.e),               # exp: t* = e
    ({"p": [0.5]}, [[0.0], [0.0], [-1.0]], [1.0, 1.0, 0.0], [-1.0], 0, 1.0),             # power: z* = 1
    ({"s": [2]}, [[0.0], [-np.sqrt(2.0)], [0.0]], [1.0, 0.0, 1.0], [1.0], 0, -1.0),      # SDP 2x2: x* = -1
    ({"bu": [0.5], "bl": [-0.5]}, [[0.0], [1.0]], [1.0, 0.5], [-1.0], 0, 1.0),           # box: x* = 1
    ({"bu": [0.35], "bl": [-0.35]}, [[0.0], [1.0]], [1.0, 0.65], [1.0], 0, 0.3),          # box: x* = 0.3
])
def test_closed_forms_per_cone(scs, cone, Adense, bvec, cvec, idx, expected):
    # R:test/test_scs_coverage.py:563-632,805-820,912-1021,1380-1410
    d = {"A": sp.csc_matrix(np.array(Adense)), "b": np.array(bvec), "c": np.array(cvec)}
    sol = scs.SCS(d, cone, verbose=False, eps_abs=1e-7, eps_rel=1e-7).solve()
    assert sol["info"]["status"] == "solved"
    assert_almost_equal(sol["x"][idx], expected, decimal=4)


def test_statuses_infeasible_and_unbounded(scs):
    # R:test/test_scs_coverage.py:862-904
    # x >= 1 and x <= 0
    d = {"A": sp.csc_matrix([[-1.0], [1.0]]), "b": np.array([-1.0, 0.0]), "c": np.array([1.0])}
    sol = scs.SCS(d, {"l": 2}, verbose=False).solve()
    assert sol["info"]["status"] == "infeasible" and sol["info"]["status_val"] == scs.INFEASIBLE
    assert np.isnan(sol["x"]).all()
    # min -x st x >= 0
    d = {"A": sp.csc_matrix([[-1.0]]), "b": np.array([0.0]), "c": np.array([-1.0])}
    sol = scs.SCS(d, {"l": 1}, verbose=False).solve()
    assert sol["info"]["status"] == "unbounded" and sol["info"]["status_val"] == scs.UNBOUNDED
    assert np.isnan(sol["y"]).all()


def test_info_dict_and_copies(scs):
    # R:test/test_scs_coverage.py:328-365,1259-1285,2865-2877,2909-2917
    solver = scs.SCS(data, {"l": 2}, verbose=False)
    sol = solver.solve()
    info = sol["info"]
    for key in ("status", "status_val", "iter", "pobj", "dobj", "gap", "res_pri", "res_dual", "res_infeas",
                "res_unbdd_a", "res_unbdd_p", "setup_time", "solve_time", "lin_sys_time", "cone_time", "accel_time",
                "scale", "comp_slack", "accepted_accel_steps", "rejected_accel_steps", "aa_stats", "scale_updates"):
        assert key in info, key
    for key in ("iter", "n_accept", "n_reject_lapack", "n_reject_rank0", "n_reject_nonfinite", "n_reject_weight_cap",
                "n_safeguard_reject", "last_rank", "last_aa_norm", "last_regularization"):
        assert key in info["aa_stats"], key
    assert isinstance(info["iter"], int) and isinstance(info["pobj"], float) and isinstance(info["status"], str)
    assert info["status_val"] == scs.SOLVED
    for k in ("setup_time", "solve_time", "lin_sys_time", "cone_time", "accel_time"):
        assert info[k] >= 0.0
    sol2 = solver.solve()
    assert sol["x"] is not sol2["x"]
    keep = sol["x"].copy()
    sol2["x"][:] = 123.0
    np.testing.assert_array_equal(sol["x"], keep)
    assert sol["x"].flags.owndata and sol["x"].shape == (1,) and sol["y"].shape == (2,)


def test_warm_start_validation_and_reuse(scs):
    # R:test/test_scs_coverage.py:2576-2598, R:test/test_scs_object.py:68-110
    dat, K, _ = helpers.load_problem("problems_rand.npz", "feas1_")
    solver = scs.SCS(dat, K, verbose=False, eps_abs=1e-7, eps_rel=1e-7)
    cold = solver.solve(warm_start=False)
    warm = solver.solve()                       # previous solution is the warm start
    assert warm["info"]["status"] == "solved" and warm["info"]["iter"] <= cold["info"]["iter"]
    again = solver.solve(warm_start=True, x=cold["x"], y=cold["y"], s=cold["s"])
    assert again["info"]["iter"] <= cold["info"]["iter"]
    with pytest.raises(ValueError):
        solver.solve(warm_start=True, x=np.zeros(3))
    with pytest.raises(ValueError):
        solver.solve(warm_start=True, y=np.zeros((len(dat["b"]), 1)))
    with pytest.raises(TypeError):
        solver.solve(warm_start=1)


def test_update_b_c(scs):
    # R:test/test_scs_object.py:68-88, R:test/test_scs_coverage.py:663-697,1141-1178,1543-1553
    dat, K, _ = helpers.load_problem("problems_rand.npz", "feas0_")
    kw = dict(verbose=False, eps_abs=1e-7, eps_rel=1e-7)
    solver = scs.SCS(dat, K, **kw)
    solver.update(b=dat["b"] * 1.0)             # allowed before the first solve
    solver.solve()
    b2, c2 = dat["b"] * 1.02, dat["c"] * 0.97
    solver.update(b=b2, c=c2)
    upd = solver.solve()
    fresh = scs.SCS(dict(dat, b=b2, c=c2), K, **kw).solve()
    assert upd["info"]["status"] == fresh["info"]["status"] == "solved"
    assert abs(upd["info"]["pobj"] - fresh["info"]["pobj"]) < 1e-5 * max(1, abs(fresh["info"]["pobj"]))
    with pytest.raises(ValueError):
        solver.update(b=np.zeros(3))
    with pytest.raises(TypeError):
        solver.update(c=[1.0, 2.0])
    with pytest.raises(TypeError):
        solver.update(b=np.arange(len(b2)))


def test_aa_off_counters_zero_and_type2(scs):
    # R:test/test_scs_coverage.py:1320-1330 and the README's type-II example (R:README.md:106-111)
    dat, K, p_star = helpers.load_problem("problems_rand.npz", "feas0_")
    info = scs.SCS(dat, K, acceleration_lookback=0, verbose=False).solve()["info"]
    assert all(v == 0 for v in info["aa_stats"].values())
    assert info["accepted_accel_steps"] == 0 and info["rejected_accel_steps"] == 0
    sol = scs.SCS(dat, K, acceleration_type_1=False, acceleration_regularization=1e-12, verbose=False,
                  eps_abs=1e-7, eps_rel=1e-7).solve()
    assert sol["info"]["status"] == "solved" and abs(sol["info"]["pobj"] - p_star) < 1e-4
    assert sol["info"]["aa_stats"]["iter"] > 0


def test_normalize_and_max_iters_and_time_limit(scs):
    dat, K, p_star = helpers.load_problem("problems_rand.npz", "feas2_")
    a = scs.SCS(dat, K, normalize=False, verbose=False, eps_abs=1e-7, eps_rel=1e-7).solve()
    assert a["info"]["status"] == "solved" and abs(a["info"]["pobj"] - p_star) < 1e-4
    few = scs.SCS(dat, K, max_iters=7, verbose=False).solve()
    assert few["info"]["iter"] == 7 and "inaccurate" in few["info"]["status"]
    tl = scs.SCS(dat, K, time_limit_secs=1e-4, eps_abs=1e-12, eps_rel=1e-12, verbose=False).solve()
    assert tl["info"]["iter"] < 10000


def test_verbose_false_prints_nothing(scs, capfd):
    # R:test/test_scs_coverage.py:2925-2929
    scs.SCS(data, {"l": 2}, verbose=False).solve()
    out = capfd.readouterr()
    assert out.out == "" and out.err == ""
    scs.SCS(data, {"l": 2}, verbose=True).solve()
    assert "status" in capfd.readouterr().out


def test_verbose_footer_reports_solution_quality_and_certificates(scs, capfd):
    """the block the reference prints under the timings (R:notebooks/scs_benchmarks.ipynb cells 2, 3 outputs): cone
    distances, complementary slackness, residuals for a solved problem; the certificate lines for an infeasible and an
    unbounded one"""
    import re
    dat, K, p_star = helpers.load_problem("problems_std.npz", "std_feas_")
    scs.SCS(dat, K, verbose=True, eps_abs=1e-6, eps_rel=1e-6).solve()
    out = capfd.readouterr().out
    m_ = re.search(r"cones: dist\(s, K\) = (\S+), dist\(y, K\*\) = (\S+)", out)
    assert m_ and float(m_.group(1)) < 1e-6 and float(m_.group(2)) < 1e-6, out[-900:]
    assert re.search(r"comp slack: s'y/\|s\|\|y\| = \S+, gap: \|x'Px\+c'x\+b'y\| = \S+", out)
    assert re.search(r"pri res: \|Ax\+s-b\| = \S+, dua res: \|Px\+A'y\+c\| = \S+", out)
    dat, K, _ = helpers.load_problem("problems_std.npz", "std_infeas_")
    scs.SCS(dat, K, verbose=True).solve()
    out = capfd.readouterr().out
    assert "status:  infeasible" in out and re.search(r"cone: dist\(y, K\*\) = \S+", out)
    m_ = re.search(r"cert: \|A'y\| = (\S+)\n\s+b'y = -1.00", out)
    assert m_ and float(m_.group(1)) < 1e-3, out[-600:]
    dat, K, _ = helpers.load_problem("problems_std.npz", "std_unbdd_")
    scs.SCS(dat, K, verbose=True).solve()
    out = capfd.readouterr().out
    assert "status:  unbounded" in out and re.search(r"cone: dist\(s, K\) = \S+", out)
    m_ = re.search(r"cert: \|Ax\+s\| = (\S+)\n\s+\|Px\| = (\S+)\n\s+c'x = -1.00", out)
    assert m_ and float(m_.group(1)) < 1e-3, out[-600:]
    assert "objective = -inf" in out


def test_independent_instances_run_concurrently(scs):
    # R:test/test_thread_safety.py:78-93 — one stream + lock per instance, GIL released in solve
    dat, K, p_star = helpers.load_problem("problems_std.npz", "std_feas_")
    ref = scs.SCS(dat, K, verbose=False).solve()
    results, errors = [None] * 6, []

    def work(i):
        try:
            results[i] = scs.SCS(dat, K, verbose=False).solve()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors
    for r in results:
        np.testing.assert_array_equal(r["x"], ref["x"])   # deterministic, also under concurrency


def test_shared_instance_is_serialised_by_its_lock(scs):
    # R:test/test_free_threading.py (shared-instance concurrent solve/update must not corrupt state)
    dat, K, _ = helpers.load_problem("problems_rand.npz", "feas0_")
    solver = scs.SCS(dat, K, verbose=False)
    base = solver.solve(warm_start=False)
    outs, errors = [], []

    def work():
        try:
            outs.append(solver.solve(warm_start=False))
            solver.update(b=dat["b"])
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors and len(outs) == 4
    for o in outs:  # (the adaptive scale persists between solves, so later trajectories differ: compare optima)
        assert o["info"]["status"] == "solved"
        assert abs(o["info"]["pobj"] - base["info"]["pobj"]) < 1e-3 * max(1.0, abs(base["info"]["pobj"]))


def _read_scship_dump(fname):
    """reader of the write_data_filename format documented in csrc/io.hpp (write_problem_data)"""
    import struct
    raw = open(fname, "rb").read()
    assert raw[:8] == b"SCSHIP01"
    pos, recs = 8, {}
    f64_tags = {2, 4, 5, 8, 9, 10, 11, 14}
    while pos < len(raw):
        tag, cnt = struct.unpack_from("<IQ", raw, pos)
        pos += 12
        dt = np.float64 if tag in f64_tags else np.int32
        recs[tag] = np.frombuffer(raw, dtype=dt, count=cnt, offset=pos).copy()
        pos += cnt * np.dtype(dt).itemsize
    return recs


def test_write_data_and_log_csv_files(scs, tmp_path):
    # R:test/test_scs_coverage.py:532-547,1728-1751,3055-3069
    dat, K, _ = helpers.load_problem("problems_rand.npz", "feas0_")
    data_file, log_file = str(tmp_path / "data.bin"), str(tmp_path / "log.csv")
    sol = scs.SCS(dat, K, verbose=False, write_data_filename=data_file, log_csv_filename=log_file).solve()
    assert sol["info"]["status"] == "solved"
    recs = _read_scship_dump(data_file)
    A = dat["A"]
    assert tuple(recs[1]) == A.shape
    np.testing.assert_array_equal(recs[11], A.data)
    np.testing.assert_array_equal(recs[12], A.indices)
    np.testing.assert_array_equal(recs[9], dat["b"])
    np.testing.assert_array_equal(recs[6], K["q"])
    lines = [ln for ln in open(log_file).read().splitlines() if ln.strip()]
    header = lines[0].split(",")
    assert header[0] == "iter" and "res_pri" in header and "diff_u_ut_nrm_2" in header and header[35] == "time"
    assert len(lines) == sol["info"]["iter"] + 2          # header + one row per iteration 0..iter
    last = lines[-1].split(",")
    assert int(last[0]) == sol["info"]["iter"]
    assert abs(float(last[header.index("res_pri")]) - sol["info"]["res_pri"]) <= 1e-12 + 1e-9 * sol["info"]["res_pri"]
    # logging must not change the answer
    plain = scs.SCS(dat, K, verbose=False).solve()
    np.testing.assert_array_equal(plain["x"], sol["x"])


def test_cs_cone_reference_cases(scs):
    """R:test/test_scs_coverage.py:2020-2055 (mixed z/l/s/cs) and :2813-2825 (standalone, b = c = identity)."""
    rng = np.random.RandomState(1234)
    cone = {"z": 1, "l": 2, "s": [3, 4], "cs": [5, 4]}
    m = 1 + 2 + 6 + 10 + 25 + 16
    P = 0.1 * sp.eye(m, format="csc")
    A = sp.random(m, m, density=0.05, format="csc", random_state=rng)
    A.data = rng.randn(A.nnz)
    data = {"P": P, "A": A, "b": rng.randn(m), "c": rng.randn(m)}
    sol = scs.SCS(data, cone, max_iters=50000, verbose=False).solve()
    assert sol["info"]["status"] in ("solved", "solved_inaccurate")
    data = {"P": 0.1 * sp.eye(4, format="csc"), "A": sp.eye(4, 4, format="csc"),
            "b": np.array([1.0, 0.0, 0.0, 1.0]), "c": np.array([1.0, 0.0, 0.0, 1.0])}
    sol = scs.solve(data, {"cs": [2]}, verbose=False)
    assert sol["info"]["status"] in ("solved", "solved_inaccurate")
    # min 0.05|x|^2 + c'x s.t. s = b - x Hermitian PSD: the unconstrained minimiser x = -10 c keeps s = 11 I > 0
    np.testing.assert_allclose(sol["x"], [-10.0, 0.0, 0.0, -10.0], atol=1e-3)
    with pytest.raises(ValueError):  # dims: cs=[2] is 4 rows, not 3
        scs.SCS({"A": sp.eye(3, 3, format="csc"), "b": np.ones(3), "c": np.ones(3)}, {"cs": [2]})


# ---- Ctrl-C: R:meson.build:118 (-DCTRLC=1), status SIGINT = -5 (R:scs/py/__init__.py:20) ----
_SIGINT_CHILD = r'''
import json, os, sys, time
sys.path[:0] = [os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np
import scs, problem_gen as pg
from scs import _scs_hip
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K = {"l": 3000, "q": [10] * 100}
def make(seed):
    data, _, _ = pg.gen_feasible(K, 1500, 8, seed, proj)
    return scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0,
                   max_iters=100000000, verbose=False)
solvers = [make(7 + i) for i in range(GROUP)]
print("READY", flush=True)
t = time.time()
sols = scs.solve_batch(solvers) if GROUP > 1 else [solvers[0].solve()]
out = [{"status": s["info"]["status"], "status_val": s["info"]["status_val"], "iter": s["info"]["iter"],
        "x_nan": bool(np.isnan(s["x"]).all() and np.isnan(s["y"]).all() and np.isnan(s["s"]).all())} for s in sols]
# the listener is gone again: Python's own handler is back (a second Ctrl-C would raise KeyboardInterrupt)
import signal
out.append({"handler_restored": signal.getsignal(signal.SIGINT) is signal.default_int_handler, "seconds": time.time() - t})
print(json.dumps(out), flush=True)
'''


@pytest.mark.parametrize("group", [1, 3], ids=["scs_solve", "solve_batch"])
def test_sigint_stops_the_device_loop_with_status_interrupted(scs, tmp_path, group):
    """A ctypes call into a device loop that would run for hours: SIGINT is caught by the library while a solve runs, the loop
    stops at the next iteration and reports status "interrupted" (-5) with NaN vectors (the reference's CTRLC=1 behaviour);
    afterwards Python's handler is in place again."""
    import json, os, signal, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sigint_child.py"
    script.write_text("ROOT = %r\nGROUP = %d\n" % (root, group) + _SIGINT_CHILD)
    p = subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        line = p.stdout.readline()
        assert line.strip() == "READY", (line, p.stderr.read() if p.poll() is not None else "")
        time.sleep(3.0)  # the solve is under way (it would not end by itself)
        assert p.poll() is None
        p.send_signal(signal.SIGINT)
        out, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 0, err[-2000:]
    res = json.loads([x for x in out.splitlines() if x.startswith("[")][-1])
    for r in res[:-1]:
        assert r["status"] == "interrupted" and r["status_val"] == -5 and r["x_nan"], r
        assert r["iter"] > 100
    assert res[-1]["handler_restored"] and res[-1]["seconds"] < 30
