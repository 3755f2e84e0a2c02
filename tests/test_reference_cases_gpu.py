"""Behaviour table of the reference's extended wrapper tests, restated for the HIP backend (run with -m gpu).

Every case below states, in this repo's own words, an expectation that the reference pins in
R:test/test_scs_coverage.py (section numbers / line ranges in the comments): accepted input formats, settings
sweeps, info-dict contract, update()/warm-start flows, per-cone closed forms and degenerate matrices.
The reference solves them with its CPU backends; here they all go through scs.SCS(...) -> scs._scs_hip.
"""
import warnings

import numpy as np
import pytest
from numpy.testing import assert_allclose
from scipy import sparse as sp

pytestmark = pytest.mark.gpu

OK = ("solved", "solved_inaccurate")


@pytest.fixture(scope="module")
def scs():
    import scs as _scs
    from scs import _scs_hip
    assert _scs_hip.device_count() > 0
    return _scs


def lp():
    """max x s.t. 0 <= x <= 1  (x* = 1): the reference's shared toy LP, R:test/test_scs_coverage.py:22-38"""
    return {"A": sp.csc_matrix(np.array([[1.0], [-1.0]])), "b": np.array([1.0, 0.0]), "c": np.array([-1.0])}


LP_CONE = {"l": 2}


def solve(scs, data, cone, **kw):
    kw.setdefault("verbose", False)
    return scs.SCS(data, cone, **kw).solve()


# ---- input formats (sections 2-4, 29, 31, 43, 52, 60: R:test/test_scs_coverage.py:116-303,1335-1360,1440-1470,2936-2960) ----
@pytest.mark.parametrize("fmt", ["csr", "coo", "lil"])
def test_non_csc_A_warns_then_solves(scs, fmt):
    d = lp()
    d["A"] = d["A"].asformat(fmt)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        solver = scs.SCS(d, LP_CONE, verbose=False)
    assert any("csc" in str(w.message).lower() for w in caught)
    assert_allclose(solver.solve()["x"], [1.0], atol=1e-2)


def test_non_csc_P_warns_then_solves(scs):
    d = lp()
    d["P"] = sp.csr_matrix(np.array([[1.0]]))
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        solver = scs.SCS(d, LP_CONE, verbose=False)
    assert any("csc" in str(w.message).lower() for w in caught)
    assert solver.solve()["info"]["status"] in OK


@pytest.mark.parametrize("which", ["b", "c", "bc"])
def test_sparse_b_c_are_flattened(scs, which):
    d = lp()
    if "b" in which:
        d["b"] = sp.csc_matrix(d["b"].reshape(-1, 1))
    if "c" in which:
        d["c"] = sp.csc_matrix(d["c"].reshape(-1, 1))
    sol = solve(scs, d, LP_CONE)
    assert sol["info"]["status"] in OK
    assert_allclose(sol["x"], [1.0], atol=1e-2)


def test_unsorted_indices_are_handled_without_touching_the_callers_matrices(scs):
    A = sp.csc_matrix((np.array([-1.0, 1.0]), np.array([1, 0]), np.array([0, 2])), shape=(2, 1))  # rows listed 1, 0
    assert not A.has_sorted_indices
    keep = A.indices.copy()
    sol = solve(scs, {"A": A, "b": np.array([1.0, 0.0]), "c": np.array([-1.0])}, LP_CONE)
    assert_allclose(sol["x"], [1.0], atol=1e-2)
    assert not A.has_sorted_indices and np.array_equal(A.indices, keep)
    # same for P (2 variables, column 0 of P stored as rows 1, 0)
    A2 = sp.block_diag([sp.csc_matrix([[1.0], [-1.0]])] * 2, format="csc")
    P = sp.csc_matrix((np.array([2.0, 1.0]), np.array([1, 0]), np.array([0, 2, 2])), shape=(2, 2))
    keep = P.indices.copy()
    sol = solve(scs, {"A": A2, "b": np.array([1.0, 0, 1, 0]), "c": np.array([-1.0, -1.0]), "P": P}, {"l": 4})
    assert sol["info"]["status"] in OK
    assert not P.has_sorted_indices and np.array_equal(P.indices, keep)


def test_P_none_full_upper_and_lower_only(scs):
    d = lp()
    x_no_p = solve(scs, d, LP_CONE)["x"]
    assert_allclose(solve(scs, dict(d, P=None), LP_CONE)["x"], x_no_p, atol=1e-4)
    # symmetric P given in full == its upper triangle (the lower entries are dropped)
    A2 = sp.block_diag([sp.csc_matrix([[1.0], [-1.0]])] * 2, format="csc")
    base = {"A": A2, "b": np.array([1.0, 0, 1, 0]), "c": np.array([-0.6, -0.4])}
    Pfull = sp.csc_matrix(np.array([[2.0, 1.0], [1.0, 2.0]]))
    xf = solve(scs, dict(base, P=Pfull), {"l": 4})["x"]
    xu = solve(scs, dict(base, P=sp.triu(Pfull, format="csc")), {"l": 4})["x"]
    assert_allclose(xf, xu, atol=1e-3)
    # P with entries only below the diagonal still solves (they are stripped: P = diag(2, 2))
    Plow = sp.csc_matrix(np.array([[2.0, 0.0], [1.0, 2.0]]))
    sol = scs.solve({"P": Plow, "A": sp.eye(2, format="csc"), "b": np.zeros(2), "c": np.ones(2)}, {"z": 2}, verbose=False)
    assert sol["info"]["status"] == "solved"
    # a structurally empty 1x1 P is the LP again
    sol = solve(scs, dict(d, P=sp.csc_matrix((1, 1))), LP_CONE)
    assert sol["info"]["status"] == "solved"
    assert_allclose(sol["x"], [1.0], atol=1e-3)


def test_dtypes_float32_accepted_integers_rejected(scs):
    d = lp()
    sol = solve(scs, {"A": d["A"], "b": d["b"].astype(np.float32), "c": d["c"].astype(np.float32)}, LP_CONE)
    assert_allclose(sol["x"], [1.0], atol=1e-2)
    sol = solve(scs, {"A": d["A"].astype(np.float32), "b": d["b"], "c": d["c"]}, LP_CONE)
    assert_allclose(sol["x"], [1.0], atol=1e-2)
    for bad in ({"b": np.array([1, 0])}, {"c": np.array([-1])}, {"A": sp.csc_matrix(np.array([[1], [-1]]))}):
        with pytest.raises((TypeError, ValueError)):
            scs.SCS(dict(d, **bad), LP_CONE, verbose=False)


# ---- settings sweeps (section 8, 45, 57-59: R:test/test_scs_coverage.py:380-527,2214-2236,2690-2800) ----
@pytest.mark.parametrize("kw", [
    dict(eps_abs=1e-9, eps_rel=1e-9), dict(eps_abs=1e-2, eps_rel=1e-2), dict(normalize=True), dict(normalize=False),
    dict(adaptive_scale=True), dict(adaptive_scale=False), dict(scale=0.5), dict(rho_x=1e-3), dict(eps_infeas=1e-4),
    dict(alpha=0.5), dict(alpha=1.0), dict(alpha=1.5), dict(alpha=1.8), dict(alpha=0.01), dict(alpha=1.99),
    dict(acceleration_lookback=0), dict(acceleration_lookback=5), dict(acceleration_lookback=10),
    dict(acceleration_lookback=5, acceleration_interval=1), dict(acceleration_lookback=5, acceleration_interval=20),
    dict(acceleration_type_1=True), dict(acceleration_type_1=False),
    dict(acceleration_relaxation=0.0), dict(acceleration_relaxation=0.5), dict(acceleration_relaxation=1.5),
    dict(acceleration_relaxation=2.0), dict(acceleration_regularization=0.0), dict(max_iters=10 ** 7),
    dict(linear_solver="hip_indirect"), dict(linear_solver="auto"),
], ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()))
def test_settings_keep_the_answer(scs, kw):
    sol = solve(scs, lp(), LP_CONE, **kw)
    assert sol["info"]["status"] in OK
    assert_allclose(sol["x"], [1.0], atol=2e-2)


def test_max_iters_one_and_tiny_time_limit_return_something(scs):
    assert solve(scs, lp(), LP_CONE, max_iters=1)["info"]["iter"] <= 1
    rng = np.random.RandomState(42)
    A = sp.random(50, 30, density=0.3, format="csc", random_state=rng)
    A.data = rng.randn(A.nnz)
    sol = solve(scs, {"A": A, "b": np.abs(rng.randn(50)) + 1.0, "c": rng.randn(30)}, {"l": 50}, time_limit_secs=1e-9)
    assert "info" in sol and sol["x"].shape == (30,)


# ---- info dict contract (sections 6, 7, 27, 28, 46, 56, 61-65: R:test/test_scs_coverage.py:310-372,1245-1330,2238-2250,2855-2917) ----
def test_info_contract(scs):
    assert (scs.SOLVED, scs.INFEASIBLE, scs.UNBOUNDED, scs.SOLVED_INACCURATE, scs.INFEASIBLE_INACCURATE,
            scs.UNBOUNDED_INACCURATE, scs.FAILED, scs.INDETERMINATE, scs.SIGINT, scs.UNFINISHED) == (1, -2, -1, 2, -7, -6, -4, -3, -5, 0)
    sol = solve(scs, lp(), LP_CONE, eps_abs=1e-8, eps_rel=1e-8)
    info = sol["info"]
    assert set(sol) >= {"x", "y", "s", "info"}
    for key in ("status", "status_val", "iter", "pobj", "dobj", "gap", "res_pri", "res_dual", "res_infeas", "res_unbdd_a",
                "res_unbdd_p", "setup_time", "solve_time", "lin_sys_time", "cone_time", "accel_time", "scale", "comp_slack",
                "accepted_accel_steps", "rejected_accel_steps", "aa_stats"):
        assert key in info, key
    for key in ("iter", "n_accept", "n_reject_lapack", "n_reject_rank0", "n_reject_nonfinite", "n_reject_weight_cap",
                "n_safeguard_reject", "last_rank", "last_aa_norm", "last_regularization"):
        assert key in info["aa_stats"], key
    assert info["status"] == "solved" and info["status_val"] == scs.SOLVED
    assert isinstance(info["iter"], int) and isinstance(info["pobj"], float) and isinstance(info["status"], str)
    assert all(info[k] >= 0.0 for k in ("setup_time", "solve_time", "lin_sys_time", "cone_time", "accel_time"))
    assert abs(info["pobj"] - info["dobj"]) < 1e-4 and abs(info["pobj"] - float(lp()["c"] @ sol["x"])) < 1e-4
    assert info["res_pri"] < 1e-4 and info["res_dual"] < 1e-4 and info["gap"] < 1e-4 and info["comp_slack"] < 1e-4
    assert info["accepted_accel_steps"] >= 0 and info["rejected_accel_steps"] >= 0
    assert sol["x"].shape == (1,) and sol["y"].shape == (2,) and sol["s"].shape == (2,)
    s2 = solve(scs, lp(), LP_CONE, adaptive_scale=False, scale=0.5)["info"]
    assert isinstance(s2["scale"], float) and np.isfinite(s2["scale"]) and s2["scale"] > 0
    assert s2.get("scale_updates", 0) == 0
    assert solve(scs, lp(), LP_CONE, adaptive_scale=True)["info"].get("scale_updates", 0) >= 0


def test_feasibility_and_complementarity_of_a_solved_lp(scs):
    d = lp()
    sol = solve(scs, d, LP_CONE, eps_abs=1e-8, eps_rel=1e-8)
    x, y, s = sol["x"], sol["y"], sol["s"]
    assert np.abs(d["A"] @ x + s - d["b"]).max() < 1e-5 and (s > -1e-6).all()     # primal
    assert np.abs(d["A"].T @ y + d["c"]).max() < 1e-5 and (y > -1e-6).all()        # dual
    assert abs(s @ y) < 1e-5


# ---- update() flows (sections 12, 25, 32, 34, 49, 55, 58: R:test/test_scs_coverage.py:657-700,1225-1240,1478-1500,1543-1553,2120-2140,2610-2660,2830-2850) ----
def test_update_flows(scs):
    solver = scs.SCS(lp(), LP_CONE, verbose=False)
    assert_allclose(solver.solve()["x"], [1.0], atol=1e-2)
    solver.update(b=np.array([2.0, 2.0]), c=np.array([1.0]))          # min x s.t. -2 <= x <= 2
    assert_allclose(solver.solve()["x"], [-2.0], atol=1e-2)
    solver.update(b=np.array([3.0, 0.0]))                              # only b: 0 <= x <= 3, still min x
    assert_allclose(solver.solve()["x"], [0.0], atol=1e-2)
    solver.update(c=np.array([-1.0]))                                  # only c: max x
    assert_allclose(solver.solve()["x"], [3.0], atol=1e-2)
    solver.update()                                                    # no-op
    assert_allclose(solver.solve()["x"], [3.0], atol=1e-2)
    for ub in (1.0, 2.0, 0.5, 3.0) * 3:                                # many update/solve cycles track
        solver.update(b=np.array([ub, 0.0]))
        sol = solver.solve()
        assert sol["info"]["status"] in OK
        assert_allclose(sol["x"], [ub], atol=1e-2)
    fresh = scs.SCS(lp(), LP_CONE, verbose=False)                      # update before the first solve
    fresh.update(b=np.array([4.0, 0.0]))
    assert_allclose(fresh.solve()["x"], [4.0], atol=1e-2)


def test_infeasible_then_update_to_feasible(scs):
    solver = scs.SCS({"A": sp.csc_matrix(np.array([[1.0], [-1.0]])), "b": np.array([-1.0, 0.0]), "c": np.array([1.0])},
                     {"l": 2}, verbose=False, max_iters=5000)
    assert solver.solve()["info"]["status"] == "infeasible"
    solver.update(b=np.array([1.0, 0.0]))
    sol = solver.solve()
    assert sol["info"]["status"] in OK
    assert_allclose(sol["x"], [0.0], atol=1e-2)


def test_qp_update_c_moves_the_optimum(scs):
    # min x^2 + c x on [0, 1]: c = -1 -> 0.5, c = -0.4 -> 0.2, c = 1 -> 0
    d = dict(lp(), P=sp.csc_matrix(np.array([[2.0]])))
    solver = scs.SCS(d, LP_CONE, verbose=False, eps_abs=1e-7, eps_rel=1e-7)
    assert_allclose(solver.solve()["x"], [0.5], atol=1e-3)
    for cval, want in ((-0.4, 0.2), (1.0, 0.0)):
        solver.update(c=np.array([cval]))
        assert_allclose(solver.solve()["x"], [want], atol=1e-3)


# ---- warm starts (sections 11, 13, 26, 40-42, 53, 54: R:test/test_scs_coverage.py:636-655,703-727,1241-1247,1905-2010,2576-2600,2662-2690,3128-3147) ----
def test_warm_start_flows(scs):
    solver = scs.SCS(lp(), LP_CONE, verbose=False, eps_abs=1e-9, eps_rel=1e-9)
    first = solver.solve(warm_start=True)                   # nothing to warm-start from: must not crash
    assert_allclose(first["x"], [1.0], atol=1e-2)
    cold = solver.solve(warm_start=False)
    warm = solver.solve(warm_start=True)
    assert warm["info"]["iter"] <= cold["info"]["iter"]
    again = solver.solve(warm_start=False)                   # cold after warm: same as the first cold solve
    assert again["info"]["iter"] == cold["info"]["iter"]
    assert_allclose(again["x"], cold["x"], atol=1e-8)
    for kw in (dict(y=cold["y"].copy()), dict(s=cold["s"].copy()), dict(x=cold["x"].copy()),
               dict(x=cold["x"].copy(), y=cold["y"].copy(), s=cold["s"].copy())):
        sol = solver.solve(warm_start=True, **kw)            # partial overrides
        assert sol["info"]["status"] in OK
        assert_allclose(sol["x"], [1.0], atol=1e-2)
    fresh = scs.SCS(lp(), LP_CONE, verbose=False)            # x, y, s on the very first solve
    sol = fresh.solve(warm_start=True, x=cold["x"], y=cold["y"], s=cold["s"])
    assert_allclose(sol["x"], [1.0], atol=1e-2)
    # legacy API: warm-start vectors travel in the data dict, all or some of them
    sol = scs.solve(dict(lp(), x=cold["x"], y=cold["y"], s=cold["s"]), LP_CONE, verbose=False)
    assert_allclose(sol["x"], [1.0], atol=1e-2)
    assert scs.solve(dict(lp(), x=cold["x"]), LP_CONE, verbose=False)["info"]["status"] in OK
    assert scs.solve(lp(), LP_CONE, verbose=False)["info"]["status"] in OK


def test_repeated_and_twin_solves_are_identical(scs):
    a = scs.SCS(lp(), LP_CONE, verbose=False)
    b = scs.SCS(lp(), LP_CONE, verbose=False)
    sa, sb = a.solve(), b.solve()
    for k in ("x", "y", "s"):
        assert np.array_equal(sa[k], sb[k])
    assert sa["info"]["iter"] == sb["info"]["iter"]
    r1, r2 = a.solve(warm_start=False), a.solve(warm_start=False)
    for k in ("x", "y", "s"):
        assert np.array_equal(r1[k], r2[k])
    sa["x"][0] = 123.0                                       # returned arrays are the caller's own copies
    assert a.solve(warm_start=False)["x"][0] != 123.0


# ---- closed forms per cone (sections 17-24, 30, 33, 35, 38, 39, 50, 74, 75: R:test/test_scs_coverage.py:816-1090,1380-1430,1503-1600,1759-1900,2803-2855) ----
def _rows(*rows):
    return sp.csc_matrix(np.array(rows, dtype=float))


SQ2 = np.sqrt(2.0)
CASES = {
    # name: (A, b, c, cone, {index: expected x})
    "zero: x = 0.7": (_rows([1]), [0.7], [-1], {"z": 1}, {0: 0.7}),
    "zero + nonneg": (_rows([1, 0], [0, -1], [0, 1]), [0.5, 0, 1], [-1, -1], {"z": 1, "l": 2}, {0: 0.5, 1: 1.0}),
    "soc: max x in the unit disc with y = 0.5^0.5 fixed": (
        _rows([0, 1], [0, 0], [-1, 0], [0, -1]), [SQ2 / 2, 1, 0, 0], [-1, 0], {"z": 1, "q": [3]}, {0: SQ2 / 2}),
    "two socs": (_rows([0, 0, 1, 0], [0, 0, 0, 1], [0, 0, -1, 0], [-1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, -1], [0, -1, 0, 0],
                       [0, 0, 0, 0]), [1, 1, 0, 0, 0.5, 0, 0, 0.3], [-1, -1, 0, 0], {"l": 2, "q": [3, 3]},
                 {0: np.sqrt(0.75), 1: np.sqrt(0.91)}),
    "sdp 2x2: min x with [[1, x], [x, 1]] psd": (_rows([0], [-SQ2], [0]), [1, 0, 1], [1], {"s": [2]}, {0: -1.0}),
    "exp: min t with (1, 1, t) in K_exp": (
        _rows([0, 1, 0], [0, 0, 1], [0, -1, 0], [0, 0, -1], [-1, 0, 0]), [1, 1, 0, 0, 0], [1, 0, 0], {"z": 2, "ep": 1}, None),
    "lp + exp: t* = e, u* = 0": (
        _rows([0, 0, 1, 0], [0, 0, 0, 1], [0, -1, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, 0, -1], [-1, 0, 0, 0]),
        [1, 1, 0, 2, 0, 0, 0], [1, 1, 0, 0], {"z": 2, "l": 2, "ep": 1}, {0: np.e, 1: 0.0}),
    "pow 0.5: max z with (1, 1, z) in K_pow": (
        _rows([0, 1, 0], [0, 0, 1], [0, -1, 0], [0, 0, -1], [-1, 0, 0]), [1, 1, 0, 0, 0], [-1, 0, 0], {"z": 2, "p": [0.5]}, {0: 1.0}),
    "two power cones": (
        _rows([0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1], [0, -1, 0, 0, 0], [0, 0, -1, 0, 0],
              [-1, 0, 0, 0, 0], [0, 0, 0, -1, 0], [0, 0, 0, 0, -1], [-1, 0, 0, 0, 0]),
        [1, 1, 1, 1, 0, 0, 0, 0, 0, 0], [-1, 0, 0, 0, 0], {"z": 4, "p": [0.5, 0.5]}, {0: 1.0}),
}


@pytest.mark.parametrize("name", list(CASES))
def test_closed_forms(scs, name):
    A, b, c, cone, want = CASES[name]
    sol = solve(scs, {"A": A, "b": np.array(b, float), "c": np.array(c, float)}, cone, eps_abs=1e-7, eps_rel=1e-7)
    assert sol["info"]["status"] in OK
    if name.startswith("exp"):  # the exp test of the reference pins the objective: t* = e (x = (t, u, v) ordering there)
        assert abs(sol["info"]["pobj"] - np.e) < 1e-3
        return
    for i, v in want.items():
        assert abs(sol["x"][i] - v) < 2e-3, (name, sol["x"])


def test_cones_that_only_need_a_sane_status(scs):
    # dual power cone p = -0.5 and a stand-alone dual exponential cone: the reference only asks for a status / not FAILED
    A = _rows([0, 1, 0], [0, 0, 1], [0, -1, 0], [0, 0, -1], [-1, 0, 0])
    sol = solve(scs, {"A": A, "b": np.array([1.0, 1, 0, 0, 0]), "c": np.array([1.0, 0, 0])}, {"z": 2, "p": [-0.5]})
    assert isinstance(sol["info"]["status"], str)
    sol = scs.solve({"A": sp.csc_matrix(np.eye(3)), "b": -np.ones(3), "c": np.ones(3)}, {"ed": 1}, verbose=False)
    assert sol["info"]["status_val"] != scs.FAILED
    sol = solve(scs, lp(), {"l": 2, "q": [], "s": [], "p": []})       # empty list fields
    assert_allclose(sol["x"], [1.0], atol=1e-2)


def test_statuses(scs):
    sol = solve(scs, {"A": sp.csc_matrix(np.array([[-1.0], [1.0]])), "b": np.array([-1.0, 0.0]), "c": np.array([1.0])},
                {"l": 2}, eps_infeas=1e-7, max_iters=10000)            # x >= 1 and x <= 0
    assert sol["info"]["status"] == "infeasible" and sol["info"]["status_val"] == scs.INFEASIBLE
    sol = solve(scs, {"A": sp.csc_matrix(np.array([[-1.0]])), "b": np.array([0.0]), "c": np.array([-1.0])}, {"l": 1},
                max_iters=10000)                                       # max x s.t. x >= 0
    assert sol["info"]["status"] == "unbounded" and sol["info"]["status_val"] == scs.UNBOUNDED
    # infeasible QP: x >= 1, x <= 0 with a quadratic objective
    sol = solve(scs, {"A": sp.csc_matrix(np.array([[-1.0], [1.0]])), "b": np.array([-1.0, 0.0]), "c": np.array([1.0]),
                      "P": sp.csc_matrix(np.array([[1.0]]))}, {"l": 2}, max_iters=10000)
    assert sol["info"]["status"] in ("infeasible", "infeasible_inaccurate")


# ---- degenerate matrices (sections 66-70: R:test/test_scs_coverage.py:2963-3050) ----
def test_degenerate_matrices(scs):
    # structurally empty A: min x s.t. s = b >= 0, x free -> unbounded (or "solved" by convention)
    sol = scs.solve({"A": sp.csc_matrix((2, 1)), "b": np.ones(2), "c": np.ones(1)}, {"l": 2}, verbose=False)
    assert sol["info"]["status_val"] in (-1, -6, 1, 2)
    # unconstrained QP behind a dummy 0 = 0 row: x* = -c / 2
    n = 5
    sol = scs.solve({"P": 2.0 * sp.eye(n, format="csc"), "A": sp.csc_matrix((1, n)), "b": np.zeros(1), "c": np.ones(n)},
                    {"z": 1}, verbose=False, eps_abs=1e-9, eps_rel=1e-9)
    assert sol["info"]["status"] == "solved"
    assert_allclose(sol["x"], -0.5 * np.ones(n), atol=1e-3)
    # every cone type at once on a random strongly convex QP
    rng = np.random.RandomState(42)
    cone = {"z": 1, "l": 2, "q": [3], "s": [2], "ep": 1, "p": [0.5]}
    A = sp.random(15, 15, density=0.1, format="csc", random_state=rng)
    A.data = rng.randn(A.nnz)
    sol = scs.solve({"P": 0.1 * sp.eye(15, format="csc"), "A": A, "b": rng.randn(15), "c": rng.randn(15)}, cone, verbose=False,
                    max_iters=50000)
    assert sol["info"]["status"] in OK


def test_box_cone_with_numpy_bounds_and_mismatch(scs):
    # 0 <= x <= 1 through the box cone with t fixed to 1: rows (t | s) = (1 | 0.5 - x), bounds -0.5 <= s <= 0.5
    A = sp.csc_matrix(np.array([[0.0], [1.0]]))
    d = {"A": A, "b": np.array([1.0, 0.5]), "c": np.array([-1.0])}
    sol = solve(scs, d, {"bu": np.array([0.5]), "bl": np.array([-0.5])})
    assert_allclose(sol["x"], [1.0], atol=1e-2)
    with pytest.raises(ValueError, match="bu different dimension"):
        scs.SCS(d, {"bu": [0.5, 0.5], "bl": [-0.5]}, verbose=False)
