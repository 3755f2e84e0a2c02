"""`LinearSolver.AUTO` resolves like the reference's (R:scs/py/__init__.py:45-54: AUTO = "the best available DIRECT solver"):
the dense direct solver of the device when the problem fits it, the indirect solver otherwise (scs/__init__.py `_resolve_auto`).
Every other test file sees AUTO pinned to the indirect module (tests/conftest.py `_auto_selects_indirect`); the tests here carry the
`auto_resolution` marker and see the real policy."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import problem_gen as pg  # noqa: E402

pytestmark = pytest.mark.auto_resolution


def _proj(z, K):
    from scs import _scs_hip
    return _scs_hip.proj_cone(z, K, dual=True)


# ------------------------------------------------------------------ policy (no GPU: the device's answer is stubbed)
def _stub_mem(monkeypatch, free):
    import scs
    monkeypatch.setattr(scs._scs_hip, "mem_info", (lambda: None) if free is None else (lambda: (int(free), int(288e9))))


def test_auto_policy_prefers_direct_when_it_fits(monkeypatch):
    import scs
    from scs import _scs_hip, _scs_hip_dense
    _stub_mem(monkeypatch, 200e9)
    A = sp.random(4050, 1350, density=0.01, format="csc", random_state=1)
    assert scs._resolve_auto(4050, 1350, A) is _scs_hip_dense
    assert scs._select_scs_module({}, 4050, 1350, A) is _scs_hip_dense
    assert scs._select_scs_module({"linear_solver": "auto"}, 4050, 1350, A) is _scs_hip_dense
    assert scs._select_scs_module({"linear_solver": scs.LinearSolver.HIP_INDIRECT}, 4050, 1350, A) is _scs_hip
    assert scs._select_scs_module({"linear_solver": "hip_indirect"}, 4050, 1350, A) is _scs_hip
    stg = {"linear_solver": scs.LinearSolver.AUTO, "verbose": False}
    scs._select_scs_module(stg, 4050, 1350, A)
    assert stg == {"verbose": False}                      # `linear_solver` never reaches the backend (R:scs/py/__init__.py:69-74)


def test_auto_policy_falls_back_to_indirect(monkeypatch):
    import scs
    from scs import _scs_hip, _scs_hip_dense
    _stub_mem(monkeypatch, 200e9)
    big = sp.random(20000, 4097, density=1e-4, format="csc", random_state=2)
    assert scs._resolve_auto(20000, 4097, big) is _scs_hip                      # beyond the measured crossover (profiles/r06_auto_crossover.txt)
    edge = sp.random(20000, 4096, density=1e-4, format="csc", random_state=2)
    assert scs._resolve_auto(20000, 4096, edge) is _scs_hip_dense
    _stub_mem(monkeypatch, 0.5e9)                                                 # 8 n^2 = 134 MB > a quarter of 0.5 GB free
    assert scs._resolve_auto(20000, 4096, edge) is _scs_hip
    _stub_mem(monkeypatch, 200e9)
    rows = sp.csc_matrix(np.ones((300, 3000)))                                   # 300 full rows: 2.7e9 products per G
    assert scs._resolve_auto(300, 3000, rows) is _scs_hip
    _stub_mem(monkeypatch, None)                                                  # no device: the indirect module reports it
    assert scs._resolve_auto(4050, 1350, None) is _scs_hip
    assert scs._resolve_auto() is _scs_hip


def test_auto_without_device_reports_like_the_indirect_backend():
    """on a box without a GPU AUTO must end in the backend's own error, not in an import or policy failure"""
    import scs
    from scs import _scs_hip
    if _scs_hip.device_count() > 0:
        pytest.skip("needs a box without a GPU")
    A = sp.eye(3, format="csc")
    with pytest.raises(ValueError, match="ScsWork allocation error"):
        scs.SCS({"A": A, "b": np.ones(3), "c": np.ones(3)}, {"l": 3}, verbose=False)


# ------------------------------------------------------------------ on the device
@pytest.mark.gpu
def test_auto_takes_the_dense_direct_solver_for_a_config5_member_and_matches_the_oracle():
    """AUTO on a config-5 member (n = 1350) reports the dense direct solver and agrees entry-wise with the
    oracle's sparse LDL' (both solve their linear systems exactly: x, y, s at rtol 1e-4)"""
    import scs
    from oracle import scs_oracle
    Kb, nb, kb, seed = pg.workload("config5_small")
    d, p_star, _ = pg.gen_feasible(Kb, nb, kb, seed + 3, _proj)
    dflt = scs.SCS(d, Kb, verbose=False).solve()                                  # no linear_solver named: AUTO
    assert dflt["info"]["status"] == "solved"
    assert dflt["info"]["lin_sys_solver"].startswith("dense-direct"), dflt["info"]["lin_sys_solver"]
    named = scs.SCS(d, Kb, verbose=False, linear_solver=scs.LinearSolver.AUTO).solve()
    assert named["info"]["lin_sys_solver"].startswith("dense-direct")
    for key in ("x", "y", "s"):
        assert np.array_equal(named[key], dflt[key])
    stg = dict(verbose=False, eps_abs=1e-10, eps_rel=1e-10, max_iters=200000)
    got = scs.SCS(d, Kb, linear_solver="auto", **stg).solve()
    ref = scs_oracle.solve(d, Kb, indirect=False, **stg)
    assert got["info"]["status"] == "solved" and ref["info"]["status"] == "solved"
    assert got["info"]["lin_sys_solver"].startswith("dense-direct")
    assert abs(got["info"]["pobj"] - p_star) <= 1e-4 * max(1.0, abs(p_star))
    for key in ("x", "y", "s"):
        np.testing.assert_allclose(got[key], ref[key], rtol=1e-4, atol=1e-4 * np.abs(ref[key]).max(), err_msg=key)
    # the explicit members still select what they name
    ind = scs.SCS(d, Kb, verbose=False, linear_solver=scs.LinearSolver.HIP_INDIRECT).solve()
    assert ind["info"]["lin_sys_solver"].startswith("sparse-indirect"), ind["info"]["lin_sys_solver"]


@pytest.mark.gpu
def test_auto_stays_indirect_beyond_the_dense_solver():
    """n = 1e5 (BASELINE config 2's shape): AUTO must not try an 80 GB inverse"""
    import scs
    K, n, k, seed = pg.workload("config2_lp_soc")
    d, p_star, _ = pg.gen_feasible(K, n, k, seed, _proj)
    sol = scs.SCS(d, K, verbose=False, max_iters=50).solve()
    assert sol["info"]["lin_sys_solver"].startswith("sparse-indirect"), sol["info"]["lin_sys_solver"]
    assert sol["info"]["iter"] == 50


@pytest.mark.gpu
def test_auto_grouped_solve_and_reference_cases_go_through_the_direct_solver():
    """the small closed-form cases of R:test/test_scs_basic.py / test_scs_coverage.py through AUTO (= direct, as in the reference's
    own runs of them), and a grouped solve of AUTO-made workspaces"""
    import scs
    data = {"A": sp.csc_matrix([1.0, -1.0]).T.tocsc(), "b": np.array([1.0, 0.0]), "c": np.array([-1.0])}
    for cone, expected in (({"q": [], "l": 2}, 1.0), ({"q": [2], "l": 0}, 0.5)):     # R:test/test_scs_basic.py:36-72 (AUTO is one of its solvers)
        sol = scs.SCS(data, cone=cone, linear_solver=scs.LinearSolver.AUTO, verbose=False).solve()
        assert sol["info"]["status"] == "solved" and sol["info"]["lin_sys_solver"].startswith("dense-direct")
        np.testing.assert_almost_equal(sol["x"][0], expected, decimal=2)
    qp = {"P": sp.csc_matrix(np.array([[3.0, -1.0], [-1.0, 2.0]])), "A": sp.csc_matrix(np.array([[-1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])),
          "b": np.array([-1.0, 0.3, -0.5]), "c": np.array([-1.0, -1.0])}          # a strictly convex QP whose bound x1 <= 0.3 is active
    s2 = scs.SCS(qp, {"z": 1, "l": 2}, verbose=False, eps_abs=1e-9, eps_rel=1e-9)
    r2 = s2.solve()
    assert r2["info"]["status"] == "solved" and r2["info"]["lin_sys_solver"].startswith("dense-direct")
    np.testing.assert_allclose(r2["x"], [0.3, -0.7], atol=1e-6)
    s2.update(b=np.array([-1.0, 0.4, -0.5]))
    np.testing.assert_allclose(s2.solve()["x"], [0.4, -0.6], atol=1e-6)
    Kb, nb, kb, seed = pg.workload("config5_small")
    probs = [pg.gen_feasible(Kb, nb, kb, seed + 40 + i, _proj)[0] for i in range(4)]
    solvers = [scs.SCS(d, Kb, verbose=False) for d in probs]
    grp = scs.solve_batch(solvers)
    assert all(r["info"]["status"] == "solved" and "dense-direct" in r["info"]["lin_sys_solver"] for r in grp)
    solo = scs.SCS(probs[2], Kb, verbose=False).solve(warm_start=False)
    for key in ("x", "y", "s"):
        assert np.array_equal(solo[key], grp[2][key])
