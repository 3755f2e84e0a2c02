"""GPU tests at BASELINE.json's full sizes, through size-independent properties (the oracle would take
hours here): adjointness and linearity of the SpMV pair, projection idempotence / Moreau identities,
and an end-to-end solve of the metric workload (m=2e6, n=1e6, nnz~2e7) checked by its KKT certificate
and by the optimum p* the instance was constructed with."""
import numpy as np
import pytest

import problem_gen as pg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from scs import _scs_hip
    assert _scs_hip.device_count() > 0
    return _scs_hip


@pytest.fixture(scope="module")
def big(hip):
    K, n, k, seed = pg.workload("target_lp_soc")
    data, p_star, xys = pg.gen_feasible(K, n, k, seed, lambda z, K: hip.proj_cone(z, K, dual=True))
    return K, data, p_star, xys


def test_spmv_adjoint_and_linear_full_size(hip, big):
    K, data, _, _ = big
    A = data["A"]
    m, n = A.shape
    rng = np.random.default_rng(0)
    x1, x2, y = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(m)
    Ax1, Ax2 = hip.spmv(A, x1), hip.spmv(A, x2)
    Aty = hip.spmv(A, y, transpose=True)
    # <A x, y> = <x, A' y>
    lhs, rhs = Ax1 @ y, x1 @ Aty
    assert abs(lhs - rhs) <= 1e-10 * (np.linalg.norm(Ax1) * np.linalg.norm(y))
    # linearity
    np.testing.assert_allclose(hip.spmv(A, 2.0 * x1 - 3.0 * x2), 2.0 * Ax1 - 3.0 * Ax2, rtol=0, atol=1e-9 * np.abs(Ax1).max())
    # against scipy on a sample of rows (different summation order => tolerance, not bits)
    ref = A @ x1
    np.testing.assert_allclose(Ax1, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    # run-to-run determinism
    np.testing.assert_array_equal(hip.spmv(A, x1), Ax1)


def test_projection_properties_full_size(hip):
    rng = np.random.default_rng(1)
    K = {"z": 100000, "l": 300000, "q": [20] * 5000, "ep": 50000, "ed": 50000,
         "p": (rng.uniform(0.1, 0.9, 33333) * rng.choice([-1.0, 1.0], 33333)).tolist()}
    m = pg.cone_dims(K)
    z = rng.standard_normal(m)
    p = hip.proj_cone(z, K)
    d = hip.proj_cone(-z, K, dual=True)
    # idempotence and Moreau: z = Pi_K(z) - Pi_K*(-z),  <Pi_K(z), Pi_K*(-z)> = 0
    np.testing.assert_allclose(hip.proj_cone(p, K), p, rtol=0, atol=1e-7)
    np.testing.assert_allclose(p - d, z, rtol=0, atol=1e-7)
    assert abs(p @ d) <= 1e-6 * max(1.0, np.linalg.norm(p) * np.linalg.norm(d))


def test_metric_workload_solves_to_certificate(hip, big):
    import scs
    K, data, p_star, (x0, y0, s0) = big
    sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=1e-5, eps_rel=1e-5, verbose=False).solve()
    info = sol["info"]
    assert info["status"] == "solved", info
    A, b, c = data["A"], data["b"], data["c"]
    x, y, s = sol["x"], sol["y"], sol["s"]
    # certificate in original units, max norm, relative as SCS's own stopping rule
    pri = np.abs(A @ x + s - b).max()
    dua = np.abs(A.T @ y + c).max()
    gap = abs(c @ x + b @ y)
    assert pri <= 1e-5 + 1e-5 * max(np.abs(A @ x).max(), np.abs(s).max(), np.abs(b).max()) * 1.01
    assert dua <= 1e-5 + 1e-5 * max(np.abs(A.T @ y).max(), np.abs(c).max()) * 1.01
    assert gap <= 1e-5 + 1e-5 * max(abs(c @ x), abs(b @ y)) * 1.01
    assert abs(info["pobj"] - p_star) <= 1e-4 * abs(p_star)
    # cone membership: s in K, y in K* (self-dual cones here)
    assert s[: K["l"]].min() >= -1e-9 and y[: K["l"]].min() >= -1e-9
    q = 10
    S = s[K["l"]:].reshape(-1, q)
    Y = y[K["l"]:].reshape(-1, q)
    assert (np.linalg.norm(S[:, 1:], axis=1) <= S[:, 0] + 1e-7).all()
    assert (np.linalg.norm(Y[:, 1:], axis=1) <= Y[:, 0] + 1e-7).all()


def test_large_qp_solves_to_certificate(hip):
    """a QP over l + q cones large enough for the column-sorted pass layouts of A, A' (split: two workgroups per row
    chunk) and P: solved, KKT certificate in original units, and the optimum the instance was constructed with"""
    import scs
    from scipy import sparse
    K = {"l": 200000, "q": [10] * 10000}
    n = 150000
    data, p_star, (x0, y0, s0) = pg.gen_feasible_qp(K, n, 20, 11, lambda z, K: hip.proj_cone(z, K, dual=True))
    assert data["A"].nnz > (1 << 20) and sparse.triu(data["P"]).nnz > 0
    sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=1e-6, eps_rel=1e-6, verbose=False).solve()
    info = sol["info"]
    assert info["status"] == "solved", info
    assert "column-sorted" in info["lin_sys_solver"]
    A, b, c = data["A"], data["b"], data["c"]
    P = data["P"]
    Pf = (P + sparse.triu(P, 1).T) if (abs(P - P.T)).nnz else P  # full symmetric P whichever triangle was passed
    x, y, s = sol["x"], sol["y"], sol["s"]
    px = Pf @ x
    pri = np.abs(A @ x + s - b).max()
    dua = np.abs(px + A.T @ y + c).max()
    gap = abs(x @ px + c @ x + b @ y)
    assert pri <= 1e-6 + 1e-6 * max(np.abs(A @ x).max(), np.abs(s).max(), np.abs(b).max()) * 1.01
    assert dua <= 1e-6 + 1e-6 * max(np.abs(px).max(), np.abs(A.T @ y).max(), np.abs(c).max()) * 1.01
    assert gap <= 1e-6 + 1e-6 * max(abs(x @ px), abs(c @ x), abs(b @ y)) * 1.01
    assert abs(info["pobj"] - p_star) <= 1e-4 * max(1.0, abs(p_star))
    np.testing.assert_allclose(x, x0, rtol=0, atol=2e-4 * max(1.0, np.abs(x0).max()))  # strictly convex: x is unique


def _certificate(data, sol, eps, P=None):
    """SCS's own stopping rule evaluated independently (scipy products, original units); returns the three margins"""
    A, b, c = data["A"], data["b"], data["c"]
    x, y, s = sol["x"], sol["y"], sol["s"]
    ax, aty = A @ x, A.T @ y
    px = P @ x if P is not None else np.zeros_like(x)
    pri = np.abs(ax + s - b).max()
    dua = np.abs(px + aty + c).max()
    gap = abs(x @ px + c @ x + b @ y)
    assert pri <= eps + eps * max(np.abs(ax).max(), np.abs(s).max(), np.abs(b).max()) * 1.01, ("pri", pri)
    assert dua <= eps + eps * max(np.abs(px).max(), np.abs(aty).max(), np.abs(c).max()) * 1.01, ("dual", dua)
    assert gap <= eps + eps * max(abs(x @ px), abs(c @ x), abs(b @ y)) * 1.01, ("gap", gap)


def test_config2_exact_workload_solves_to_certificate(hip):
    """BASELINE.json configs[1]: random LP+SOC m=2e5 n=1e5 nnz~2e6, indirect CG, AA lookback 10 — the exact workload"""
    import scs
    K, n, k, seed = pg.workload("config2_lp_soc")
    data, p_star, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: hip.proj_cone(z, K, dual=True))
    assert data["A"].shape == (200000, 100000)
    sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=1e-6, eps_rel=1e-6, verbose=False,
                  acceleration_lookback=10).solve()
    info = sol["info"]
    assert info["status"] == "solved", info
    assert info["aa_stats"]["n_accept"] > 0
    _certificate(data, sol, 1e-6)
    assert abs(info["pobj"] - p_star) <= 1e-5 * abs(p_star)
    s, y = sol["s"], sol["y"]
    assert s[: K["l"]].min() >= -1e-9 and y[: K["l"]].min() >= -1e-9
    S, Y = s[K["l"]:].reshape(-1, 10), y[K["l"]:].reshape(-1, 10)
    assert (np.linalg.norm(S[:, 1:], axis=1) <= S[:, 0] + 1e-7).all() and (np.linalg.norm(Y[:, 1:], axis=1) <= Y[:, 0] + 1e-7).all()


def _config3_cone():
    return pg.workload("config3_mixed")[0]  # BASELINE.json configs[2], the 99,999-bound box cone included


def test_config3_projections_full_size_with_box(hip):
    """every cone type of config 3 at its size, INCLUDING the 99,999-bound box cone: idempotence and Moreau"""
    K = _config3_cone()
    m = pg.cone_dims(K)
    assert m == 999999
    rng = np.random.default_rng(7)
    z = rng.standard_normal(m)
    p = hip.proj_cone(z, K)
    d = hip.proj_cone(-z, K, dual=True)
    np.testing.assert_allclose(hip.proj_cone(p, K), p, rtol=0, atol=1e-7)
    np.testing.assert_allclose(p - d, z, rtol=0, atol=1e-7)
    assert abs(p @ d) <= 1e-6 * max(1.0, np.linalg.norm(p) * np.linalg.norm(d))
    # box block: t >= 0 and bl t <= s <= bu t
    o = K["z"] + K["l"]
    t, sb = p[o], p[o + 1:o + 100000]
    assert t >= 0 and (sb <= np.array(K["bu"]) * t + 1e-9).all() and (sb >= np.array(K["bl"]) * t - 1e-9).all()


def test_config3_mixed_cones_with_box_solves_to_certificate(hip):
    """BASELINE.json configs[2]: mixed z/l/box/q/exp/pow, m~1e6 n=5e5 nnz~1e7 — a FULL solve, checked by its KKT
    certificate in original units, the constructed optimum p*, and cone membership of s and y by the cones' defining inequalities in numpy"""
    import scs
    K = _config3_cone()
    data, p_star, _ = pg.gen_feasible(K, 500000, 20, 3, lambda z, K: hip.proj_cone(z, K, dual=True))
    assert data["A"].nnz > 9e6
    sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=1e-4, eps_rel=1e-4, verbose=False).solve()
    info = sol["info"]
    assert info["status"] == "solved", info
    _certificate(data, sol, 1e-4)
    assert abs(info["pobj"] - p_star) <= 1e-3 * max(1.0, abs(p_star))
    # membership of s in K and of y in K* by the cones' defining inequalities, evaluated in numpy (helpers.cone_violation) — the HIP
    # projection is not its own judge here (R:test/test_solve_random_cone_prob.py:55-65)
    import helpers
    s, y = sol["s"], sol["y"]
    for vec, dual in ((s, False), (y, True)):
        viol = helpers.cone_violation(vec, K, dual=dual)
        assert set(viol) == {"z", "l", "box", "q", "ep", "ed", "p"}
        tol = 1e-5 * max(1.0, np.abs(vec).max())
        assert max(viol.values()) <= tol, (dual, viol)
    assert abs(s @ y) <= 1e-4 * max(1.0, np.linalg.norm(s) * np.linalg.norm(y))  # complementary slackness


def test_config4_psd_heavy_solves_to_certificate(hip):
    """BASELINE.json configs[3]: s=[200]*50 + l — the batched MFMA eigen-solve path (split mode, warm starts) in a
    full solve: certificate, p*, and PSD membership of every 200 x 200 block of s and y by LAPACK"""
    import scs
    import helpers
    K = {"l": 1000, "s": [200] * 50}
    m = pg.cone_dims(K)
    assert m == 1006000
    data, p_star, _ = pg.gen_feasible(K, 335000, 30, 4, lambda z, K: hip.proj_cone(z, K, dual=True))
    sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=1e-4, eps_rel=1e-4, verbose=False).solve()
    info = sol["info"]
    assert info["status"] == "solved", info
    _certificate(data, sol, 1e-4)
    assert abs(info["pobj"] - p_star) <= 1e-4 * max(1.0, abs(p_star))  # the solver's own tolerance
    o, d = K["l"], 200 * 201 // 2
    for vec in (sol["s"], sol["y"]):
        assert vec[: K["l"]].min() >= -1e-8
        scale = max(1.0, np.abs(vec).max())
        for i in range(50):
            assert np.linalg.eigvalsh(helpers.svec_to_sym(vec[o + i * d:o + (i + 1) * d], 200)).min() > -1e-6 * scale


def test_config4_shaped_qp_matches_constructed_solution_entrywise(hip):
    """BASELINE.json configs[3] says "compare x/y to CPU within 1e-4".  A conic LP of this shape has a non-unique dual
    (test above: certificate + p*), so this is the strictly convex QP of the same shape — s = [200] * 50 + l,
    m = 1 006 000 — whose (x, y, s) is unique and known by construction (LAPACK eigh projections, independent of the
    oracle and of the HIP kernels): entry-wise agreement at 1e-4 of the largest entry, through the split-mode MFMA
    eigen-solve path with warm starts.  Solved at eps 1e-8: at 1e-7 the dual's entry-wise error is 0.7e-4 .. 1.6e-4 of
    the largest entry (the accuracy eps buys, measured with either PSD stopping level — tools/dbg/config4_qp_entry.py),
    i.e. on the bar itself; at 1e-8 it is 5e-6"""
    import scs
    import helpers
    K = {"l": 1000, "s": [200] * 50}
    data, p_star, (x0, y0, s0) = pg.gen_feasible_qp(K, 335000, 30, 44, helpers.proj_dual_l_s_numpy)
    assert data["A"].shape == (1006000, 335000)
    sol = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=1e-8, eps_rel=1e-8, verbose=False,
                  max_iters=20000).solve()
    info = sol["info"]
    assert info["status"] == "solved", info
    assert abs(info["pobj"] - p_star) <= 1e-6 * max(1.0, abs(p_star))
    for key, ref in (("x", x0), ("y", y0), ("s", s0)):
        np.testing.assert_allclose(sol[key], ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max(), err_msg=key)


def test_power_law_rows_at_the_metric_size_iterate_as_the_csr_stream_layout(hip, monkeypatch):
    """the metric workload's size with Pareto(1.3) row lengths (up to 20 000 nonzeros per row): the rows too long for the pass layout's
    6-bit count fields ride in the passes as pieces (8 rows per lane, ~80 000 piece slots) and every fused CG epilogue / residual product
    runs on them through the piece-sum launch.  The instance converges slowly (a solve to 1e-4 needs far more than 1500 iterations:
    tools/dbg/powerlaw_solve.py), so this checks 12 plain ADMM iterations (no acceleration: nothing chaotic) against the same
    iterations on the plain CSR-stream layout (9419 row blocks: the size at which the residual epilogues' partials overran their buffer
    until round 3) — same iterates to the accuracy of the inexact linear solves, same residuals, CG step counts within 3 %."""
    import scs
    K, n, k, seed = pg.workload("powerlaw_lp")
    data, p_star, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: hip.proj_cone(z, K, dual=True), pattern=pg.workload_pattern("powerlaw_lp"))
    assert np.diff(data["A"].tocsr().indptr).max() > 5000
    stg = dict(linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, max_iters=12, acceleration_lookback=0,
               verbose=False)
    got = scs.SCS(data, K, **stg).solve()
    assert "long rows in pieces" in got["info"]["lin_sys_solver"], got["info"]["lin_sys_solver"]
    monkeypatch.setenv("SCS_HIP_SLAB", "0")
    ref = scs.SCS(data, K, **stg).solve()
    assert "CSR-stream" in ref["info"]["lin_sys_solver"]
    assert got["info"]["iter"] == ref["info"]["iter"] == 12
    # (12 iterations: this far from the solution the iteration amplifies rounding — where an inexact CG solve stops flips with the order of
    # the row sums — and after 150 iterations ALL four layouts have drifted apart by per cents: tools/dbg/powerlaw_layouts.py)
    assert abs(got["info"]["cg_iters"] - ref["info"]["cg_iters"]) <= 2
    for key in ("x", "y", "s"):
        np.testing.assert_allclose(got[key], ref[key], rtol=0, atol=1e-6 * np.abs(ref[key]).max(), err_msg=key)
    for key in ("res_pri", "res_dual", "pobj"):
        assert abs(got["info"][key] - ref["info"][key]) <= 1e-5 * abs(ref["info"][key]) + 1e-9, (key, got["info"][key], ref["info"][key])


@pytest.mark.labs
def test_cg_with_the_dot_product_in_k1_iterates_as_the_one_with_it_in_k2(hip, monkeypatch):
    """round 4 (csrc/cg_k1dot.hpp): on large LPs / SOCPs p'Gp is formed as sum (A p)_i z_i + sum r_x p_j^2 — K1's epilogue and the kernel
    that forms p — and K2 stores raw A'z.  Same mathematics, another fixed summation order: on BASELINE config 2 (m = 2e5, both matrices on
    the column-sorted layout) the first 60 iterations with either formulation (SCS_HIP_K1DOT=0: the dot in K2) agree to 1e-9 of the largest
    entry and take the same number of CG steps to within a step per linear solve; run-ahead and synchronous loops of the new one agree bit for bit."""
    K, n, k, seed = pg.workload("config2_lp_soc")
    data, _, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: hip.proj_cone(z, K, dual=True))
    import helpers
    args = helpers.raw_args(data, K)
    stg = dict(eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False, max_iters=60)
    out = {}
    for tag, env in (("k2", {"SCS_HIP_K1DOT": "0"}), ("k1", {"SCS_HIP_K1DOT": "1"}), ("k1_sync", {"SCS_HIP_K1DOT": "1", "SCS_HIP_PIPELINE": "0", "SCS_HIP_GRAPH": "0"})):
        for kk in ("SCS_HIP_K1DOT", "SCS_HIP_PIPELINE", "SCS_HIP_GRAPH"):
            monkeypatch.delenv(kk, raising=False)
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        out[tag] = hip.SCS(*args, **stg).solve(False, None, None, None)
        assert "column-sorted" in out[tag]["info"]["lin_sys_solver"]
    for key in ("x", "y", "s"):
        np.testing.assert_array_equal(out["k1"][key], out["k1_sync"][key], err_msg=key)
        scl = np.abs(out["k2"][key]).max()
        np.testing.assert_allclose(out["k1"][key], out["k2"][key], rtol=0, atol=1e-9 * scl, err_msg=key)
    assert out["k1"]["info"]["cg_iters"] == out["k1_sync"]["info"]["cg_iters"]
    assert abs(out["k1"]["info"]["cg_iters"] - out["k2"]["info"]["cg_iters"]) <= 60
