"""CPU tests (no GPU): the ORACLE is pinned against the reference's own fixtures.

 * cone projections  vs golden vectors captured from R:test/gen_random_cone_prob.py
 * generated problems vs their certified optimum p* (same fixtures the reference's certificate
   tests draw: R:test/test_solve_random_cone_prob.py:46-91, R:test/test_scs_rand.py:92-126,
   R:test/test_scs_sdp.py:92-126)
 * closed-form known answers restated from R:test/test_scs_coverage.py and R:test/test_scs_basic.py
"""
import numpy as np
import pytest
from scipy import sparse

import helpers
import problem_gen as pg
from oracle import scs_oracle as oracle

TIGHT = dict(eps_abs=1e-7, eps_rel=1e-7, eps_infeas=1e-7, verbose=False)


@pytest.mark.parametrize("case", helpers.load_projection_cases(), ids=lambda c: c[0])
def test_projection_goldens(case):
    tag, K, z, gproj, gdual = case
    if z.size == 0:
        return
    exp_like = K["ep"] + K["ed"] > 0
    powlike = len(K["p"]) > 0
    # SOC/PSD/l/z are closed form: 1e-9.  The reference's Python power-cone Newton stops at |f|<1e-8
    # (R:test/gen_random_cone_prob.py:177-203) and its exp bisection at 1e-9 on rho with an inner Newton
    # tolerance of 1e-6 (:259-313): goldens are only that accurate (the oracle is checked separately
    # against the projection optimality conditions in test_exp_cone_optimality).
    atol = 1e-4 if exp_like else (1e-6 if powlike else 1e-9)
    scl = max(1.0, np.abs(z).max())
    np.testing.assert_allclose(oracle.proj_cone(z, K), gproj, rtol=0, atol=atol * scl)
    np.testing.assert_allclose(oracle.proj_cone(z, K, dual=True), gdual, rtol=0, atol=atol * scl)


def test_exp_cone_optimality():
    """Moreau: v = Pi_K(v) + Pi_K°(v), <Pi_K(v), Pi_K°(v)> = 0, Pi_K(v) in K, -Pi_K°(v) in K*."""
    rng = np.random.RandomState(0)
    for scl in (0.01, 1.0, 100.0):
        V = scl * rng.randn(300, 3)
        P = oracle.proj_cone(V.ravel(), {"ep": 300}).reshape(-1, 3)
        D = V - P  # polar part; -D must be in the dual cone
        # the root search stops on ABSOLUTE tolerances (1e-8 heuristics, 1e-15 on h): ~1e-8 |v| in <P, D>
        assert np.abs((P * D).sum(1)).max() <= 1e-8 * scl * max(1.0, scl)
        r, s, t = P.T
        inside = (s > 0) & (s * np.exp(np.minimum(r / np.where(s > 0, s, 1), 700)) <= t * (1 + 1e-9) + 1e-12)
        boundary = (s <= 1e-12 * scl) & (r <= 1e-12 * scl) & (t >= -1e-12 * scl)
        assert (inside | boundary).all()
        Pd = oracle.proj_cone((-V).ravel(), {"ed": 300}).reshape(-1, 3)
        np.testing.assert_allclose(Pd, -D, atol=1e-8 * max(1.0, scl))


def test_box_cone_projection_is_a_minimiser():
    rng = np.random.RandomState(1)
    bl, bu = -rng.rand(20) - 0.1, rng.rand(20) + 0.1
    K = {"bu": bu.tolist(), "bl": bl.tolist()}
    for _ in range(20):
        z = 3 * rng.randn(21)
        p = oracle.proj_cone(z, K)
        t, x = p[0], p[1:]
        assert t >= 0 and (x <= t * bu + 1e-9).all() and (x >= t * bl - 1e-9).all()
        d0 = np.sum((p - z) ** 2)
        for tt in np.linspace(max(t - 0.5, 0), t + 0.5, 21):  # 1-D convex in t
            q = np.concatenate([[tt], np.clip(z[1:], tt * bl, tt * bu)])
            assert np.sum((q - z) ** 2) >= d0 - 1e-9


@pytest.mark.parametrize("fname,prefix,decimal", [
    ("problems_std.npz", "std_feas_", 3),
    ("problems_rand.npz", "feas0_", 2), ("problems_rand.npz", "feas1_", 2), ("problems_rand.npz", "feas2_", 2),
    ("problems_sdp.npz", "feas0_", 2), ("problems_sdp.npz", "feas1_", 2), ("problems_sdp.npz", "feas2_", 2),
])
@pytest.mark.parametrize("indirect", [False, True], ids=["ldl", "cg"])
def test_feasible_certificates(fname, prefix, decimal, indirect):
    data, K, p_star = helpers.load_problem(fname, prefix)
    sol = oracle.solve(data, K, indirect=indirect, **TIGHT)
    assert sol["info"]["status"] == "solved"
    x, y, s = sol["x"], sol["y"], sol["s"]
    np.testing.assert_almost_equal(data["c"] @ x, p_star, decimal=decimal)
    np.testing.assert_almost_equal(-data["b"] @ y, p_star, decimal=decimal)
    assert np.linalg.norm(data["A"] @ x - data["b"] + s) < 1e-3
    assert np.linalg.norm(data["A"].T @ y + data["c"]) < 1e-3
    np.testing.assert_almost_equal(s @ y, 0.0, decimal=5)
    np.testing.assert_almost_equal(s, oracle.proj_cone(s, K), decimal=4)
    np.testing.assert_almost_equal(y, oracle.proj_cone(y, K, dual=True), decimal=3)


@pytest.mark.parametrize("fname,prefix", [("problems_std.npz", "std_infeas_"), ("problems_rand.npz", "infeas0_"),
                                          ("problems_rand.npz", "infeas1_"), ("problems_sdp.npz", "infeas0_")])
@pytest.mark.parametrize("indirect", [False, True], ids=["ldl", "cg"])
def test_infeasible(fname, prefix, indirect):
    data, K, _ = helpers.load_problem(fname, prefix)
    sol = oracle.solve(data, K, indirect=indirect, **dict(TIGHT, eps_abs=1e-5, eps_rel=1e-5, eps_infeas=1e-5))
    assert sol["info"]["status"] == "infeasible"
    y = sol["y"]
    assert np.linalg.norm(data["A"].T @ y) < 1e-3 and data["b"] @ y < -0.1


@pytest.mark.parametrize("fname,prefix", [("problems_std.npz", "std_unbdd_"), ("problems_rand.npz", "unbdd0_"),
                                          ("problems_sdp.npz", "unbdd1_")])
def test_unbounded_direct(fname, prefix):
    # the reference only runs this class on its direct backend (R:test/test_solve_random_cone_prob.py:79-91)
    data, K, _ = helpers.load_problem(fname, prefix)
    sol = oracle.solve(data, K, indirect=False, **dict(TIGHT, eps_abs=1e-5, eps_rel=1e-5, eps_infeas=1e-5))
    assert sol["info"]["status"] == "unbounded"
    x, s = sol["x"], sol["s"]
    assert np.linalg.norm(data["A"] @ x + s) < 1e-3 and data["c"] @ x < -0.1


def test_config1_lp_known_optimum():
    """BASELINE.json configs[0]: LP m=4000 n=2000 nnz~1e5; p* also cross-checked with HiGHS."""
    data, K, p_star = helpers.load_problem("problem_config1_lp.npz", "lp_")
    sol = oracle.solve(data, K, indirect=True, eps_abs=1e-6, eps_rel=1e-6, verbose=False)
    assert sol["info"]["status"] == "solved"
    assert abs(sol["info"]["pobj"] - p_star) < 1e-3 * abs(p_star)


# ---------------------------------------------------------------- closed forms (R:test/test_scs_coverage.py)
def _tiny(A, b, c, P=None):
    d = {"A": sparse.csc_matrix(np.atleast_2d(A)), "b": np.asarray(b, float), "c": np.asarray(c, float)}
    if P is not None:
        d["P"] = sparse.csc_matrix(np.atleast_2d(P))
    return d


@pytest.mark.parametrize("indirect", [False, True], ids=["ldl", "cg"])
def test_closed_forms(indirect):
    kw = dict(indirect=indirect, verbose=False, eps_abs=1e-7, eps_rel=1e-7)
    # min -x st x<=1, x>=0  -> x*=1        (R:test/test_scs_basic.py:35-72)
    d = _tiny([[1.0], [-1.0]], [1.0, 0.0], [-1.0])
    assert abs(oracle.solve(d, {"l": 2}, **kw)["x"][0] - 1.0) < 1e-4
    # same data over one SOC of dim 2: (1 - x, x) in Q2 -> x* = 0.5
    assert abs(oracle.solve(d, {"q": [2]}, **kw)["x"][0] - 0.5) < 1e-4
    # QP: min 0.5 x^2 - x st 0<=x<=0.5 ... unconstrained minimiser 1 clipped by x<=0.5
    d = _tiny([[1.0], [-1.0]], [0.5, 0.0], [-1.0], P=[[1.0]])
    assert abs(oracle.solve(d, {"l": 2}, **kw)["x"][0] - 0.5) < 1e-4
    # zero cone: x = 0.7
    d = _tiny([[1.0]], [0.7], [1.0])
    assert abs(oracle.solve(d, {"z": 1}, **kw)["x"][0] - 0.7) < 1e-5
    # exp cone: min t st (1, 1, t) in K_exp  -> t* = e      (R:test/test_scs_coverage.py:912-951)
    d = _tiny([[0.0], [0.0], [-1.0]], [1.0, 1.0, 0.0], [1.0])
    assert abs(oracle.solve(d, {"ep": 1}, **kw)["x"][0] - np.e) < 1e-4
    # power cone: max z st (1,1,z) in K_0.5 -> z*=1        (R:test/test_scs_coverage.py:984-1021)
    d = _tiny([[0.0], [0.0], [-1.0]], [1.0, 1.0, 0.0], [-1.0])
    assert abs(oracle.solve(d, {"p": [0.5]}, **kw)["x"][0] - 1.0) < 1e-4
    # SDP 2x2: min x st [[1, x],[x, 1]] >= 0 -> x* = -1     (R:test/test_scs_coverage.py:1380-1410)
    d = _tiny([[0.0], [-np.sqrt(2.0)], [0.0]], [1.0, 0.0, 1.0], [1.0])
    assert abs(oracle.solve(d, {"s": [2]}, **kw)["x"][0] + 1.0) < 1e-4
    # box cone (t, s): row 0 has A=0, b=1 so s0 = t = 1      (R:test/test_scs_coverage.py:563-632)
    #   max x st 0<=x<=1: s1 = 0.5 - x in [-0.5, 0.5] -> x* = 1
    d = _tiny([[0.0], [1.0]], [1.0, 0.5], [-1.0])
    sol = oracle.solve(d, {"bu": [0.5], "bl": [-0.5]}, **kw)
    assert sol["info"]["status"] == "solved" and abs(sol["x"][0] - 1.0) < 1e-4
    #   min x st 0.3<=x<=1 -> x* = 0.3
    d = _tiny([[0.0], [1.0]], [1.0, 0.65], [1.0])
    assert abs(oracle.solve(d, {"bu": [0.35], "bl": [-0.35]}, **kw)["x"][0] - 0.3) < 1e-4
    #   max x1+x2 st 0<=x1<=1, -1<=x2<=1 -> (1, 1)
    d = _tiny([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]], [1.0, 0.5, 0.0], [-1.0, -1.0])
    x = oracle.solve(d, {"bu": [0.5, 1.0], "bl": [-0.5, -1.0]}, **kw)["x"]
    assert abs(x[0] - 1.0) < 1e-4 and abs(x[1] - 1.0) < 1e-4


def test_direct_vs_indirect_agree():
    # R:test/test_scs_coverage.py:2060-2080: backends agree to 4 dp at eps = 1e-9
    data, K, _ = helpers.load_problem("problems_std.npz", "std_feas_")
    a = oracle.solve(data, K, indirect=False, eps_abs=1e-9, eps_rel=1e-9, verbose=False)
    b = oracle.solve(data, K, indirect=True, eps_abs=1e-9, eps_rel=1e-9, verbose=False)
    np.testing.assert_almost_equal(a["x"], b["x"], decimal=4)
    np.testing.assert_almost_equal(a["s"], b["s"], decimal=4)


def test_normalize_on_off_same_answer():
    # R:test/test_scs_coverage.py:2189-2195
    data, K, p_star = helpers.load_problem("problems_rand.npz", "feas0_")
    a = oracle.solve(data, K, normalize=True, **TIGHT)
    b = oracle.solve(data, K, normalize=False, **TIGHT)
    assert abs(a["info"]["pobj"] - p_star) < 1e-4 and abs(b["info"]["pobj"] - p_star) < 1e-4


def test_warm_start_update_and_determinism():
    data, K, _ = helpers.load_problem("problems_rand.npz", "feas1_")
    args = helpers.raw_args(data, K)
    s1 = oracle.OracleSCS(*args, **TIGHT)
    r1 = s1.solve(False)
    r2 = s1.solve(True)  # warm start from the previous solution: far fewer iterations
    assert r2["info"]["status"] == "solved" and r2["info"]["iter"] <= r1["info"]["iter"]
    s2 = oracle.OracleSCS(*args, **TIGHT)
    np.testing.assert_array_equal(s2.solve(False)["x"], r1["x"])  # bit-determinism across instances
    b2 = data["b"] * 1.01
    s1.update(b=b2)
    r3 = s1.solve(True)
    fresh = oracle.solve(dict(data, b=b2), K, **TIGHT)
    assert r3["info"]["status"] == fresh["info"]["status"] == "solved"
    assert abs(r3["info"]["pobj"] - fresh["info"]["pobj"]) < 1e-4 * max(1, abs(fresh["info"]["pobj"]))


def test_aa_off_counters_zero():
    # R:test/test_scs_coverage.py:1320-1330
    data, K, _ = helpers.load_problem("problems_rand.npz", "feas0_")
    info = oracle.solve(data, K, acceleration_lookback=0, verbose=False)["info"]
    assert all(v == 0 for v in info["aa_stats"].values())
    assert info["accepted_accel_steps"] == 0 and info["rejected_accel_steps"] == 0


def test_embedded_qp_warm_start_regression():
    # R:test/test_warm_start_consistency.py:257-301 — cold / warm x2 must all be 'solved'
    d = np.load(helpers.GOLDEN + "/warm_start_qp.npz")
    P = sparse.csc_matrix((d["P_data"], d["P_indices"], d["P_indptr"]), shape=(15, 15))
    G = sparse.csc_matrix((d["G_data"], d["G_indices"], d["G_indptr"]), shape=(60, 15))
    data = {"P": P, "A": G, "b": d["h"].copy(), "c": d["q"].copy()}
    kw = dict(verbose=False, normalize=True, max_iters=100000, scale=0.1, adaptive_scale=True, eps_abs=1e-7,
              eps_rel=1e-6, eps_infeas=1e-7, alpha=1.5, rho_x=1e-6, acceleration_interval=10)
    for lb in (0, 10):
        args = helpers.raw_args(data, {"l": 60})
        s = oracle.OracleSCS(*args, acceleration_lookback=lb, **kw)
        w1 = s.solve(True, d["x0"].copy(), d["y0"].copy(), d["s0"].copy())
        w2 = s.solve(True, d["x0"].copy(), d["y0"].copy(), d["s0"].copy())
        c = oracle.OracleSCS(*args, acceleration_lookback=lb, **kw).solve(False)
        assert c["info"]["status"] == w1["info"]["status"] == w2["info"]["status"] == "solved"


# ---- complex PSD cone `cs` (R:test/test_spectral_and_complex_cones.py:121-152,
# R:test/test_mix_sd_csd_cone.py:31-40, R:test/test_scs_coverage.py:2028-2034,2822) -----------------------------
@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 8, 13])
def test_cs_projection_vs_complex_eigh(k):
    rng = np.random.RandomState(40 + k)
    for scl in (1.0, 30.0):
        v = scl * rng.randn(k * k)
        ref = helpers.proj_hermitian_psd(v, k)
        for dual in (False, True):  # self-dual
            got = oracle.proj_cone(v, {"cs": [k]}, dual=dual)
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * scl * max(k, 1))
    # idempotent, and the residual v - Pi(v) is in -K and orthogonal to Pi(v) (Moreau)
    if k:
        v = rng.randn(k * k)
        pv = oracle.proj_cone(v, {"cs": [k]})
        np.testing.assert_allclose(oracle.proj_cone(pv, {"cs": [k]}), pv, atol=1e-12)
        assert abs(pv @ (v - pv)) < 1e-12 * k * k
        assert np.linalg.eigvalsh(helpers.cvec_to_herm(pv - v, k)).min() > -1e-12


def _cs_qp(cone, seed, density, p_scale):
    """the reference's own instance construction for the cs tests (R:test/test_spectral_and_complex_cones.py:55-71)"""
    rng = np.random.RandomState(seed)
    m = pg.cone_dims(cone)
    n = m
    P = p_scale * sparse.eye(n, format="csc")
    A = sparse.random(m, n, density=density, format="csc", random_state=rng)
    A.data = rng.randn(A.nnz)
    c = rng.randn(n)
    b = A @ rng.randn(n) + np.abs(rng.randn(m))
    return dict(P=P, A=A, b=b, c=c)


@pytest.mark.parametrize("cone,seed", [({"cs": [3]}, 42), ({"cs": [2, 3]}, 123), (dict(z=1, l=2, s=[3], cs=[3]), 456),
                                       (dict(z=1, l=2, s=[3, 4], cs=[5, 4]), 1234)])
def test_cs_solves_reference_cases(cone, seed):
    data = _cs_qp(cone, seed, 0.5, 1.0)
    sol = oracle.solve(data, cone, eps_abs=1e-7, eps_rel=1e-7)
    assert sol["info"]["status"] == "solved"
    pri, dual, gap = helpers.kkt_certificate(data, sol, P=data["P"])
    assert pri < 1e-5 and dual < 1e-5 and gap < 1e-5
    # s in K, y in K* = K: Hermitian blocks PSD
    o = pg.cone_dims({k: v for k, v in cone.items() if k != "cs"})
    for k in cone["cs"]:
        for vec in (sol["s"], sol["y"]):
            assert np.linalg.eigvalsh(helpers.cvec_to_herm(vec[o:o + k * k], k)).min() > -1e-6
        o += k * k


def test_numpy_cone_membership_agrees_with_the_oracle_projections():
    """helpers.cone_violation (the defining inequalities of every cone family and of its dual, used by the full-size GPU tests as
    the judge of membership) against the oracle: what the oracle projects onto K / K* is inside to 1e-8, a random point is not."""
    import helpers
    import problem_gen as pg
    rng = np.random.RandomState(1)
    K = {"z": 3, "l": 5, "bu": [1.0, 2.0, 0.5], "bl": [-1.0, 0.0, -3.0], "q": [4, 3], "s": [3, 2], "ep": 6, "ed": 5,
         "p": [0.3, -0.6, 0.5, -0.25]}
    m = pg.cone_dims(K)
    for dual in (False, True):
        for _ in range(10):
            z = 3.0 * rng.randn(m)
            viol = helpers.cone_violation(oracle.proj_cone(z, K, dual=dual), K, dual=dual)
            assert set(viol) == {"z", "l", "box", "q", "s", "ep", "ed", "p"}
            assert max(viol.values()) <= 1e-8, (dual, viol)
            outside = helpers.cone_violation(z, K, dual=dual)
            assert sum(x > 1e-3 for x in outside.values()) >= 5, outside


def test_symbolic_ldl_entry_matches_the_factorisation():
    """oracle/oscs_linsys.c o_lin_sys_symbolic (the fill table of DESIGN §7, tools/ldl_fill_table.py): nnz(L) of the symbolic phase alone is
    what the full direct backend allocates for the same pattern; the elimination tree of a banded KKT pattern is one chain"""
    import problem_gen as pg
    from oracle import scs_oracle
    K = {"l": 300}
    data, _, _ = pg.gen_feasible(K, 100, 6, 3, lambda z, KK: scs_oracle.proj_cone(z, KK, dual=True))
    lnz, height = scs_oracle.ldl_symbolic(data["A"])
    info = scs_oracle.solve(data, K, indirect=False, verbose=False, max_iters=5)["info"]
    assert "nnz(L)=%d" % lnz in info["lin_sys_solver"], (lnz, info["lin_sys_solver"])
    assert 1 <= height <= 400 and lnz >= data["A"].nnz
    band = pg.banded_sparse(400, 200, 5, np.random.default_rng(0))
    lnz_b, height_b = scs_oracle.ldl_symbolic(band)
    assert lnz_b < 20 * 600 and height_b > 500   # fill proportional to N, a tree that is (almost) one chain
