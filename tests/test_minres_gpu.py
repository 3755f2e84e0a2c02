"""MINRES on the KKT system with the zero-cone block un-eliminated (csrc/minres.hpp) — the second Krylov method of the indirect
linear solve (north_star "CG/MINRES over A and A'"; the reference's indirect backends: R:meson.build:261,303-304).

Checked against the oracle's direct LDL' solve of the same problems (entry-wise where the solution is unique), against the PCG
path of the same library, and for the properties the reference's tests pin: determinism
(R:test/test_scs_coverage.py:2283-2301), warm start / update (R:test/test_scs_object.py:68-88)."""
import numpy as np
import pytest

import helpers
import problem_gen as pg

pytestmark = [pytest.mark.gpu, pytest.mark.labs]  # MINRES lost its measurement (profiles/r05_config3_minres.txt): it lives in the -DSCS_HIP_LABS build

STG = dict(eps_abs=1e-9, eps_rel=1e-9, eps_infeas=1e-9, verbose=False, max_iters=20000)


@pytest.fixture(scope="module")
def hip():
    from scs import _scs_hip
    assert _scs_hip.device_count() > 0, "GPU tests need a HIP device (no CPU fallback exists)"
    return _scs_hip


@pytest.fixture(scope="module")
def oracle():
    from oracle import scs_oracle
    return scs_oracle


def _xys(got, ref, rtol=1e-4):
    for key in ("x", "y", "s"):
        np.testing.assert_allclose(got[key], ref[key], rtol=rtol, atol=rtol * np.abs(ref[key]).max(), err_msg=key)


CASES = {
    # zero-cone heavy LP + SOC: 30 % equality rows
    "z_lp_soc": ({"z": 120, "l": 200, "q": [12, 7, 30]}, 150, 8, False),
    # every cone family behind a zero cone (the shape of BASELINE config 3, oracle-friendly size)
    "z_mixed": ({"z": 60, "l": 150, "bu": [1.0, 2.0, 0.5, 4.0], "bl": [-1.0, 0.0, -3.0, -0.1], "q": [9, 5], "s": [6, 4], "ep": 7, "ed": 5,
                 "p": [0.3, -0.6, 0.5]}, 110, 7, False),
    # strictly convex QP (P != 0: the (1,1) block carries P), unique x, y, s
    "z_qp": ({"z": 80, "l": 100, "q": [10, 6]}, 260, 6, True),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_minres_solves_match_the_oracle_and_pcg(hip, oracle, monkeypatch, case):
    K, n, k, qp = CASES[case]
    proj = lambda z, K: oracle.proj_cone(z, K, dual=True)
    if qp:
        data, p_star, (x0, y0, s0) = pg.gen_feasible_qp(K, n, k, 31, proj)
    else:
        data, p_star, (x0, y0, s0) = pg.gen_feasible(K, n, k, 31, proj)
    args = helpers.raw_args(data, K)
    sols = {}
    for mode in ("cg", "minres"):
        monkeypatch.setenv("SCS_HIP_KRYLOV", mode)
        sols[mode] = hip.SCS(*args, **STG).solve(False, None, None, None)
        assert sols[mode]["info"]["status"] == "solved", (mode, sols[mode]["info"])
        assert abs(sols[mode]["info"]["pobj"] - p_star) < 1e-6 * max(1.0, abs(p_star)), mode
    assert "MINRES" in sols["minres"]["info"]["lin_sys_solver"] and "MINRES" not in sols["cg"]["info"]["lin_sys_solver"]
    ref = oracle.OracleSCS(*args, indirect=False, **STG).solve(False)
    assert ref["info"]["status"] == "solved"
    if qp:  # unique: entry-wise against the oracle's direct solve and against the constructed optimum
        _xys(sols["minres"], ref)
        _xys(sols["minres"], {"x": x0, "y": y0, "s": s0})
    cert_m = helpers.kkt_certificate(data, sols["minres"], P=data.get("P"))
    cert_c = helpers.kkt_certificate(data, sols["cg"], P=data.get("P"))
    for a, b in zip(cert_m, cert_c):  # primal / dual residual, gap in original units: as good as what PCG ends with
        assert a < max(1e-6, 10.0 * b), (cert_m, cert_c)
    # (the two methods stop on the same reduced residual but leave different error: the ADMM paths — and iteration counts — differ;
    #  what must agree is where they end: the optimal value and the certificate above; entry-wise where the solution is unique)
    if qp:
        _xys(sols["minres"], sols["cg"], rtol=1e-5)


def test_minres_certificates_of_infeasible_and_unbounded_problems(hip, oracle, monkeypatch):
    """problems with a zero cone whose iterates diverge towards a certificate: the linear solves must stay accurate relative to
    a growing iterate (the reference's statuses: R:test/test_scs_coverage.py:862-904)"""
    monkeypatch.setenv("SCS_HIP_KRYLOV", "minres")
    from scipy import sparse
    rng = np.random.RandomState(4)
    # infeasible: x1 + x2 = 1 (z), x <= -1 componentwise (l)
    A = sparse.csc_matrix(np.vstack([np.ones((1, 2)), np.eye(2)]))
    data = {"A": A, "b": np.array([1.0, -1.0, -1.0]), "c": rng.randn(2)}
    sol = hip.SCS(*helpers.raw_args(data, {"z": 1, "l": 2}), eps_abs=1e-7, eps_rel=1e-7, eps_infeas=1e-8, verbose=False, max_iters=5000).solve(False, None, None, None)
    assert sol["info"]["status"] == "infeasible", sol["info"]
    # unbounded: minimise -x1 subject to x1 - x2 = 0 (z), x2 >= 0
    A = sparse.csc_matrix(np.array([[1.0, -1.0], [0.0, -1.0]]))
    data = {"A": A, "b": np.zeros(2), "c": np.array([-1.0, 0.0])}
    sol = hip.SCS(*helpers.raw_args(data, {"z": 1, "l": 1}), eps_abs=1e-7, eps_rel=1e-7, eps_infeas=1e-8, verbose=False, max_iters=5000).solve(False, None, None, None)
    assert sol["info"]["status"] == "unbounded", sol["info"]


def test_minres_is_deterministic_and_the_run_ahead_loop_keeps_its_bits(hip, oracle, monkeypatch):
    K, n, k, _ = CASES["z_lp_soc"]
    data, _, _ = pg.gen_feasible(K, n, k, 5, lambda z, K: oracle.proj_cone(z, K, dual=True))
    args = helpers.raw_args(data, K)
    monkeypatch.setenv("SCS_HIP_KRYLOV", "minres")
    stg = dict(STG)
    stg.update(eps_abs=1e-7, eps_rel=1e-7, max_iters=60)  # (bits, not convergence: every iteration of the last variant stalls ~100 times)
    a = hip.SCS(*args, **stg).solve(False, None, None, None)
    b = hip.SCS(*args, **stg).solve(False, None, None, None)
    monkeypatch.setenv("SCS_HIP_PIPELINE", "0")  # one host look per iteration instead of whole iterations queued ahead
    c = hip.SCS(*args, **stg).solve(False, None, None, None)
    monkeypatch.setenv("SCS_HIP_PIPELINE", "3")  # run-ahead with chunks of 3 CG steps, too short on purpose: every iteration stalls and is finished synchronously
    d = hip.SCS(*args, **stg).solve(False, None, None, None)
    for other in (b, c, d):
        assert other["info"]["iter"] == a["info"]["iter"] and other["info"]["cg_iters"] == a["info"]["cg_iters"]
        for key in ("x", "y", "s"):
            np.testing.assert_array_equal(other[key], a[key], err_msg=key)


def test_minres_warm_start_and_update(hip, oracle, monkeypatch):
    monkeypatch.setenv("SCS_HIP_KRYLOV", "minres")
    K, n, k, _ = CASES["z_qp"]
    data, p_star, _ = pg.gen_feasible_qp(K, n, k, 8, lambda z, K: oracle.proj_cone(z, K, dual=True))
    s = hip.SCS(*helpers.raw_args(data, K), **STG)
    cold = s.solve(False, None, None, None)
    assert cold["info"]["status"] == "solved"
    warm = s.solve(True, cold["x"], cold["y"], cold["s"])
    assert warm["info"]["status"] == "solved" and warm["info"]["iter"] <= max(25, cold["info"]["iter"] // 4)
    b2 = data["b"] * 1.01
    s.update(b2, None)
    upd = s.solve(True, None, None, None)
    ref = oracle.OracleSCS(*helpers.raw_args(dict(data, b=b2), K), indirect=False, **STG).solve(False)
    assert upd["info"]["status"] == "solved" and ref["info"]["status"] == "solved"
    _xys(upd, ref)


def test_default_is_pcg_and_auto_mode_switches_only_where_pcg_is_slow(hip, oracle, monkeypatch):
    """SCS_HIP_KRYLOV unset: PCG (MINRES lost the whole-solve comparison on config 3: profiles/r05_config3_minres.txt).
    SCS_HIP_KRYLOV=auto: PCG unless the cone has >= 256 zero rows and PCG needed ~100 steps per solve"""
    monkeypatch.delenv("SCS_HIP_KRYLOV", raising=False)
    K, n, k, _ = CASES["z_lp_soc"]
    data, _, _ = pg.gen_feasible(K, n, k, 5, lambda z, K: oracle.proj_cone(z, K, dual=True))
    sol = hip.SCS(*helpers.raw_args(data, K), **STG).solve(False, None, None, None)
    assert "MINRES" not in sol["info"]["lin_sys_solver"]
    monkeypatch.setenv("SCS_HIP_KRYLOV", "auto")
    sol = hip.SCS(*helpers.raw_args(data, K), **STG).solve(False, None, None, None)
    assert "MINRES" not in sol["info"]["lin_sys_solver"]  # 120 zero rows: below the threshold of the auto mode
    assert sol["info"]["status"] == "solved"


def test_auto_mode_never_switches_inside_a_queued_linear_solve(hip, oracle, monkeypatch):
    """with SCS_HIP_KRYLOV=auto the switch to MINRES used to be decided while iteration i + 1 was enqueued behind a
    still-queued iteration i; if i then stalled it was finished with MINRES steps that never had their start (1 / beta = inf, NaNs).
    Now the decision is only taken on an empty queue.  Forced here: the shape of `z_lp_soc` scaled to 300 zero-cone rows (auto asks for
    >= 256), a scale that makes the PCG solves long (auto asks for ~100 steps) and CG chunks of 3 steps (EVERY queued iteration stalls).
    (A problem whose zero-cone block is nearly square — z = 300, n = 320 — makes MINRES itself break down from the first iteration on,
    with or without stalls: tools/dbg/mr_auto_stall.py.  One more reason it lives in the labs build.)"""
    K = {"z": 300, "l": 500, "q": [30, 18, 75]}
    data, p_star, _ = pg.gen_feasible(K, 375, 8, 31, lambda z, K: oracle.proj_cone(z, K, dual=True))
    args = helpers.raw_args(data, K)
    stg = dict(STG)
    stg.update(max_iters=300, scale=25.0, adaptive_scale=False)   # (scale 25: ~115 PCG steps per solve on this problem, so auto does switch)
    sols = {}
    for pipe in ("1", "3", "0"):
        monkeypatch.setenv("SCS_HIP_KRYLOV", "auto")
        monkeypatch.setenv("SCS_HIP_PIPELINE", pipe)
        sols[pipe] = hip.SCS(*args, **stg).solve(False, None, None, None)
    ref = sols["1"]
    assert "MINRES" in ref["info"]["lin_sys_solver"], ref["info"]["lin_sys_solver"]   # the switch happened
    if not all(np.all(np.isfinite(ref[k])) for k in "xys"):
        pytest.skip("MINRES itself broke down on this problem without any stall (labs experiment): nothing to compare")
    # forced stalls (chunks of 3) and the synchronous loop do not change what the auto mode computes: the same decisions, taken on an
    # empty queue, the same bits — and no NaN from a recurrence that never had its start
    for pipe in ("3", "0"):
        other = sols[pipe]
        assert all(np.all(np.isfinite(other[k])) for k in "xys"), (pipe, other["info"])
        assert other["info"]["lin_sys_solver"] == ref["info"]["lin_sys_solver"]
        assert other["info"]["iter"] == ref["info"]["iter"] and other["info"]["cg_iters"] == ref["info"]["cg_iters"]
        for key in ("x", "y", "s"):
            np.testing.assert_array_equal(other[key], ref[key], err_msg="pipeline %s: %s" % (pipe, key))
