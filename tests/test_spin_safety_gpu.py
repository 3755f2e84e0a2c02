"""Concurrency contract of the kernels that spin on inter-workgroup barriers (k_psd_sweep_mc; csrc/psd.hpp, work.hpp SpinChain).

R:test/test_thread_safety.py:78-93 asks that independent SCS instances solve truly concurrently (GIL released,
R:scs/scsobject.h:984-987, one lock per instance).  A PSD cone of order >= 64 in a small batch spreads the Jacobi sweeps of
one matrix over several CUs with spinning barriers and a grid sized to the WHOLE device — two such grids from two streams
must never wait for each other's workgroups."""
import threading

import numpy as np
import pytest

import helpers
import problem_gen as pg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def problem():
    K = {"l": 100, "s": [200] * 8}  # 8 matrices on 256 CUs: psd_mc_members > 1 (multi-CU sweeps)
    m = pg.cone_dims(K)
    data, p_star, _ = pg.gen_feasible(K, m // 3, 20, 77, helpers.proj_dual_l_s_numpy)
    return data, K, p_star


def _solve(data, K, out, key, **kw):
    import scs
    try:
        out[key] = scs.SCS(data, K, verbose=False, eps_abs=1e-5, eps_rel=1e-5, max_iters=3000, **kw).solve()
    except Exception as e:  # surfaces in the asserting thread
        out[key] = e


def test_concurrent_instances_with_multi_cu_psd_sweeps(problem):
    from scs import _scs_hip
    data, K, p_star = problem
    out = {}
    _solve(data, K, out, "solo")
    solo = out["solo"]
    assert not isinstance(solo, Exception), solo
    assert solo["info"]["status"] == "solved", solo["info"]
    assert abs(solo["info"]["pobj"] - p_star) < 1e-3 * max(1.0, abs(p_star))
    before = _scs_hip.spin_fallbacks()
    for rep in range(2):
        threads = [threading.Thread(target=_solve, args=(data, K, out, "t%d" % i)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
            assert not t.is_alive(), "a concurrent solve hangs"
        for i in range(4):
            got = out["t%d" % i]
            assert not isinstance(got, Exception), got
            assert got["info"]["status"] == "solved", got["info"]
            assert got["info"]["iter"] == solo["info"]["iter"]
            for key in ("x", "y", "s"):  # the bits of the solo solve
                np.testing.assert_array_equal(got[key], solo[key], err_msg="thread %d %s" % (i, key))
    assert _scs_hip.spin_fallbacks() == before, "the chain of spinning launches should have kept the barriers from timing out"


def test_forced_barrier_timeout_restarts_the_solve_without_spinning_kernels(problem, monkeypatch):
    """SCS_HIP_SPIN_BUDGET_LOG2=0: a member gives up at the second poll of a barrier — what another process on the same GPU would
    cause after seconds.  The solve must still end `solved`, with the bits of a solve that never used the multi-CU kernel."""
    from scs import _scs_hip
    data, K, _ = problem
    out = {}
    monkeypatch.setenv("SCS_HIP_PSD_MC", "1")  # one workgroup per matrix from the start
    _solve(data, K, out, "ref")
    monkeypatch.delenv("SCS_HIP_PSD_MC")
    monkeypatch.setenv("SCS_HIP_SPIN_BUDGET_LOG2", "0")
    before = _scs_hip.spin_fallbacks()
    _solve(data, K, out, "got")
    ref, got = out["ref"], out["got"]
    assert not isinstance(ref, Exception) and not isinstance(got, Exception), (ref, got)
    assert _scs_hip.spin_fallbacks() == before + 1, "the barrier was expected to time out with a budget of one poll"
    assert got["info"]["status"] == ref["info"]["status"] == "solved", got["info"]
    assert got["info"]["iter"] == ref["info"]["iter"]
    for key in ("x", "y", "s"):
        np.testing.assert_array_equal(got[key], ref[key], err_msg=key)
