"""The reference's concurrency suites on the HIP backend: every class of R:test/test_free_threading.py:90-985 and the four tests of
R:test/test_thread_safety.py:32-118, scenario by scenario (same problems, same thread counts unless a comment says otherwise), plus
what only a device backend can get wrong: more live workspaces than the stream pool holds, the dense / grouped /
indirect paths driven from different threads at once, a failing `scs_init` next to running solves, a handle destroyed under its
users, a solve cut short by `time_limit_secs` while another thread updates.

Contract under test (R:scs/scsobject.h:892-905,939-945,984-987,1210-1246): one lock per instance, released on every
error path; independent instances truly concurrent (one HIP stream each while the pool lasts, shared streams beyond); results are
fresh copies.  Where the answer is deterministic the threads' results are compared BIT FOR BIT with a solve made alone
(R:test/test_scs_coverage.py:2283-2301 asks that of two sequential solves; concurrency must not change it)."""
import gc
import threading
from concurrent.futures import ThreadPoolExecutor, as_completed

import numpy as np
import pytest
import scipy.sparse as sp
from numpy.testing import assert_almost_equal

import problem_gen as pg

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(120)]

NUM_THREADS = 8  # R:test/test_free_threading.py:87


@pytest.fixture(scope="module")
def scs():
    import scs as _scs
    from scs import _scs_hip
    assert _scs_hip.device_count() > 0, "GPU tests need a HIP device (no CPU fallback exists)"
    return _scs


# ---------------------------------------------------------------- the reference's problems (R:test/test_free_threading.py:24-71)
def _make_simple_lp():
    """max x s.t. 0 <= x <= 1: x = 1"""
    A = sp.csc_matrix([1.0, -1.0]).T.tocsc()
    return {"A": A, "b": np.array([1.0, 0.0]), "c": np.array([-1.0])}, {"l": 2}, 1.0


def _make_socp():
    """max x s.t. (1 - x, x) in SOC: x = 0.5"""
    A = sp.csc_matrix([1.0, -1.0]).T.tocsc()
    return {"A": A, "b": np.array([1.0, 0.0]), "c": np.array([-1.0])}, {"q": [2]}, 0.5


def _make_larger_lp(n=20, seed=42):
    rng = np.random.RandomState(seed)
    m = 3 * n
    A_dense = rng.randn(m, n)
    x_feas = np.abs(rng.randn(n)) + 0.1
    b_ineq = A_dense @ x_feas + np.abs(rng.randn(m)) + 0.1
    A_full = sp.vstack([sp.csc_matrix(A_dense), sp.eye(n, format="csc") * -1.0], format="csc")
    return {"A": A_full, "b": np.concatenate([b_ineq, np.zeros(n)]), "c": -np.abs(rng.randn(n))}, {"l": m + n}


def _run_threads(targets, timeout=60):
    """start one thread per callable, join them all, fail on a thread that does not come back (a lock that was not released)"""
    errors, lock = [], threading.Lock()

    def wrap(fn):
        def run():
            try:
                fn()
            except Exception as e:  # noqa: BLE001 — the reference collects every exception the same way (:441-448)
                with lock:
                    errors.append("%s: %r" % (getattr(fn, "__name__", "worker"), e))
        return run

    threads = [threading.Thread(target=wrap(t)) for t in targets]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=timeout)
        assert not t.is_alive(), "thread did not complete — a lock not released on some path?"
    assert not errors, errors


def _same_bits(a, b, what=""):
    for key in ("x", "y", "s"):
        np.testing.assert_array_equal(a[key], b[key], err_msg="%s %s" % (what, key))
    assert a["info"]["iter"] == b["info"]["iter"], what


# ================================================================ R:test/test_free_threading.py:90-230
class TestConcurrentIndependentInstances:
    def test_concurrent_lp_and_socp_solves(self, scs):
        """:98-132 — and, beyond the reference: every thread gets the BITS of a solve made alone"""
        for make in (_make_simple_lp, _make_socp):
            data, cone, expected = make()
            solo = scs.SCS(data, cone, verbose=False).solve()

            def worker():
                sol = scs.SCS(data, cone, verbose=False).solve()
                assert sol["info"]["status_val"] == 1
                assert_almost_equal(sol["x"][0], expected, decimal=2)
                return sol

            with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
                results = [f.result(timeout=30) for f in [pool.submit(worker) for _ in range(NUM_THREADS)]]
            assert len(results) == NUM_THREADS
            for r in results:
                _same_bits(r, solo, make.__name__)

    def test_concurrent_mixed_problems(self, scs):
        """:134-162"""
        probs = [_make_simple_lp(), _make_socp()]

        def worker(i):
            data, cone, expected = probs[i % 2]
            sol = scs.SCS(data, cone, verbose=False).solve()
            assert sol["info"]["status_val"] == 1
            assert_almost_equal(sol["x"][0], expected, decimal=2)
            return i

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            assert sorted(f.result(timeout=30) for f in [pool.submit(worker, i) for i in range(NUM_THREADS)]) == list(range(NUM_THREADS))

    def test_concurrent_direct_and_indirect(self, scs):
        """:164-184 — the reference alternates QDLDL and CPU_INDIRECT; here the device's direct and indirect solvers"""
        data, cone, expected = _make_simple_lp()
        backends = [scs.LinearSolver.HIP_DENSE, scs.LinearSolver.HIP_INDIRECT]

        def worker(ls):
            sol = scs.SCS(data, cone, linear_solver=ls, verbose=False).solve()
            assert sol["info"]["status_val"] == 1
            assert_almost_equal(sol["x"][0], expected, decimal=2)
            assert sol["info"]["lin_sys_solver"].startswith("dense-direct" if ls is scs.LinearSolver.HIP_DENSE else "sparse-indirect")
            return True

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            assert all(f.result(timeout=30) for f in [pool.submit(worker, backends[i % 2]) for i in range(NUM_THREADS)])

    def test_concurrent_larger_problems(self, scs):
        """:186-202"""
        data, cone = _make_larger_lp(n=20)
        solo = scs.SCS(data, cone, verbose=False, max_iters=5000).solve()
        assert solo["info"]["status_val"] in (1, 2)

        def worker():
            return scs.SCS(data, cone, verbose=False, max_iters=5000).solve()

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            for f in [pool.submit(worker) for _ in range(NUM_THREADS)]:
                _same_bits(f.result(timeout=60), solo, "larger lp")

    def test_many_sequential_solves_per_thread(self, scs):
        """:204-230 (10 fresh instances per thread)"""
        data, cone, expected = _make_simple_lp()

        def worker():
            out = []
            for _ in range(10):
                sol = scs.SCS(data, cone, verbose=False).solve()
                assert sol["info"]["status_val"] == 1
                out.append(sol["x"][0])
            return out

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            allr = [f.result(timeout=60) for f in [pool.submit(worker) for _ in range(NUM_THREADS)]]
        assert all(len(r) == 10 for r in allr)
        for r in allr:
            for v in r:
                assert_almost_equal(v, expected, decimal=2)


# ================================================================ R:test/test_free_threading.py:233-280, R:test/test_thread_safety.py:39-76
class TestConcurrentSharedInstance:
    def test_shared_instance_concurrent_and_repeated_solve(self, scs):
        data, cone, expected = _make_simple_lp()
        solver = scs.SCS(data, cone, verbose=False)

        def worker():
            sol = solver.solve()
            assert sol["info"]["status_val"] == 1
            return sol["x"][0]

        for _ in range(5):  # (:260-280: five rounds of four concurrent solves, warm-started by whoever came before)
            with ThreadPoolExecutor(max_workers=4) as pool:
                for f in [pool.submit(worker) for _ in range(4)]:
                    assert_almost_equal(f.result(timeout=30), expected, decimal=2)


# ================================================================ R:test/test_free_threading.py:283-405, R:test/test_thread_safety.py:95-118
class TestConcurrentSolveUpdate:
    def test_shared_instance_concurrent_solve_and_update(self, scs):
        """:287-335"""
        data, cone, _ = _make_simple_lp()
        solver = scs.SCS(data, cone, verbose=False)
        barrier = threading.Barrier(NUM_THREADS)

        def solve_worker():
            barrier.wait(timeout=10)
            assert solver.solve()["info"]["status_val"] in (1, 2)

        def update_worker():
            barrier.wait(timeout=10)
            solver.update(b=np.array([1.0, 0.0]))

        _run_threads([solve_worker if i % 2 == 0 else update_worker for i in range(NUM_THREADS)], timeout=30)
        assert solver.solve()["info"]["status_val"] in (1, 2)  # still usable after the barrage

    def test_concurrent_update_sequences(self, scs):
        """:337-368 (own instance per thread: solve -> update c -> solve -> update b -> solve)"""
        data, cone, _ = _make_simple_lp()

        def worker():
            solver = scs.SCS(data, cone, verbose=False)
            sol1 = solver.solve()
            assert sol1["info"]["status_val"] == 1
            assert_almost_equal(sol1["x"][0], 1.0, decimal=2)
            solver.update(c=np.array([1.0]))
            sol2 = solver.solve()
            assert sol2["info"]["status_val"] == 1
            assert_almost_equal(sol2["x"][0], 0.0, decimal=2)
            solver.update(b=np.array([1.0, 1.0]))
            sol3 = solver.solve()
            assert sol3["info"]["status_val"] == 1
            assert_almost_equal(sol3["x"][0], -1.0, decimal=2)
            return True

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            assert all(f.result(timeout=60) for f in [pool.submit(worker) for _ in range(NUM_THREADS)])

    def test_concurrent_warm_start(self, scs):
        """:370-405"""
        data, cone, expected = _make_simple_lp()

        def worker():
            solver = scs.SCS(data, cone, verbose=False)
            sol1 = solver.solve()
            assert_almost_equal(sol1["x"][0], expected, decimal=2)
            sol2 = solver.solve()
            assert_almost_equal(sol2["x"][0], expected, decimal=2)
            assert sol2["info"]["iter"] <= sol1["info"]["iter"]
            sol3 = solver.solve(x=np.array([0.9]), y=sol2["y"], s=sol2["s"])
            assert_almost_equal(sol3["x"][0], expected, decimal=2)
            return True

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            assert all(f.result(timeout=60) for f in [pool.submit(worker) for _ in range(NUM_THREADS)])


# ================================================================ R:test/test_free_threading.py:408-430
class TestConcurrentLegacySolve:
    def test_concurrent_legacy_solve(self, scs):
        data, cone, expected = _make_simple_lp()

        def worker():
            sol = scs.solve(data, cone, verbose=False)
            assert sol["info"]["status_val"] == 1
            assert_almost_equal(sol["x"][0], expected, decimal=2)
            return True

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            assert all(f.result(timeout=30) for f in [pool.submit(worker) for _ in range(NUM_THREADS)])


# ================================================================ R:test/test_free_threading.py:433-487
class TestThreadStress:
    def test_rapid_thread_creation(self, scs):
        """:437-462 — 50 short-lived threads, each a fresh instance (more than the 32 streams the pool creates)"""
        data, cone, expected = _make_simple_lp()

        def worker():
            sol = scs.SCS(data, cone, verbose=False).solve()
            assert_almost_equal(sol["x"][0], expected, decimal=2)

        _run_threads([worker] * 50, timeout=30)

    def test_concurrent_construction_and_solve(self, scs):
        """:464-487"""
        data, cone, expected = _make_simple_lp()
        barrier = threading.Barrier(NUM_THREADS)

        def worker():
            barrier.wait(timeout=10)
            sol = scs.SCS(data, cone, verbose=False).solve()
            assert sol["info"]["status_val"] == 1
            assert_almost_equal(sol["x"][0], expected, decimal=2)
            return True

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            assert all(f.result(timeout=30) for f in [pool.submit(worker) for _ in range(NUM_THREADS)])


# ================================================================ R:test/test_free_threading.py:490-557
class TestResultIsolation:
    def test_result_vectors_are_independent(self, scs):
        """:494-520 — and the arrays a solve hands out are not views of the instance's buffers: the next solve must not touch them"""
        data, cone, expected = _make_simple_lp()
        keep, lock = [], threading.Lock()

        def worker(tid):
            solver = scs.SCS(data, cone, verbose=False)
            sol = solver.solve()
            snap = {k: sol[k].copy() for k in "xys"}
            solver.update(c=np.array([1.0]))
            other = solver.solve()  # a different answer in the instance's own buffers
            assert abs(other["x"][0]) < 0.05
            for k in "xys":
                np.testing.assert_array_equal(sol[k], snap[k])
                assert sol[k].flags.owndata and not np.shares_memory(sol[k], other[k])
            with lock:
                keep.append(snap)

        _run_threads([lambda i=i: worker(i) for i in range(NUM_THREADS)], timeout=30)
        assert len(keep) == NUM_THREADS
        for r in keep:
            assert_almost_equal(r["x"][0], expected, decimal=2)

    def test_different_problems_correct_results(self, scs):
        """:522-557"""
        probs = {"lp": _make_simple_lp(), "socp": _make_socp()}

        def worker(kind):
            data, cone, expected = probs[kind]
            sol = scs.SCS(data, cone, verbose=False).solve()
            assert_almost_equal(sol["x"][0], expected, decimal=2)
            return kind, sol["x"][0]

        with ThreadPoolExecutor(max_workers=NUM_THREADS) as pool:
            futs = [pool.submit(worker, "lp" if i % 2 == 0 else "socp") for i in range(NUM_THREADS)]
            for f in as_completed(futs, timeout=30):
                kind, val = f.result()
                assert_almost_equal(val, probs[kind][2], decimal=2)


# ================================================================ R:test/test_free_threading.py:561-683
class TestSharedConeContainers:
    """the reference's TestBorrowedRefSafety: 20 threads build instances from ONE cone dict — int value (:572-605), list value
    (:607-644), float-list value (:646-683).  The Python layer here copies every cone field into its own array before the core sees it."""

    def test_shared_cone_dict_concurrent_init(self, scs):
        data, cone, expected = _make_simple_lp()

        def worker():
            sol = scs.SCS(data, cone, verbose=False).solve()
            assert sol["info"]["status_val"] == 1
            assert_almost_equal(sol["x"][0], expected, decimal=2)

        _run_threads([worker] * 20, timeout=30)
        assert cone == {"l": 2}

    def test_shared_cone_with_list_values_concurrent_init(self, scs):
        A = sp.vstack([-sp.eye(3, n=2, format="csc"), sp.eye(2, format="csc")], format="csc")
        data = {"A": A, "b": np.array([1.0, 0.0, 0.0, 0.0, 0.0]), "c": np.array([-1.0, 0.0])}
        cone = {"q": [3], "l": 2}

        def worker():
            assert scs.SCS(data, cone, verbose=False).solve()["info"]["status_val"] in (1, 2)

        _run_threads([worker] * 20, timeout=30)
        assert cone == {"q": [3], "l": 2}

    def test_shared_cone_with_float_list_values_concurrent_init(self, scs):
        data = {"A": -sp.eye(3, n=3, format="csc"), "b": np.array([1.0, 1.0, 0.0]), "c": np.array([0.0, 0.0, -1.0])}
        cone = {"p": [0.5]}

        def worker():
            assert scs.SCS(data, cone, verbose=False, max_iters=10000).solve()["info"]["status_val"] in (1, 2, -1, -2, -7)

        _run_threads([worker] * 20, timeout=30)


# ================================================================ R:test/test_free_threading.py:686-743
class TestTOCTOURaceSafety:
    def test_rapid_create_solve_destroy(self, scs):
        """:695-727 — 8 threads x 25 create / solve / destroy cycles (the reference runs 50; a cycle is ~10 ms of launches here)"""
        data, cone, expected = _make_simple_lp()

        def worker():
            for _ in range(25):
                solver = scs.SCS(data, cone, verbose=False)
                sol = solver.solve()
                assert sol["info"]["status_val"] == 1
                assert_almost_equal(sol["x"][0], expected, decimal=2)
                del solver
            gc.collect()

        _run_threads([worker] * NUM_THREADS, timeout=90)

    def test_handle_destroyed_under_its_users(self, scs):
        """the race the class is named after (:687-693: solve / update check `self->work` — a concurrent dealloc may clear it between
        the check and the lock): here the check IS under the lock, so a thread that finishes the workspace while others solve and
        update makes them raise "Workspace not initialized!" (R:scs/scsobject.h:921-924) — never a crash, never a hang"""
        data, cone, expected = _make_simple_lp()
        for _ in range(5):
            solver = scs.SCS(data, cone, verbose=False)
            raw = solver._solver
            barrier = threading.Barrier(5)
            seen = []

            def user(op):
                barrier.wait(timeout=10)
                for _ in range(6):
                    try:
                        if op == "solve":
                            sol = solver.solve()
                            assert sol["info"]["status_val"] == 1
                            assert_almost_equal(sol["x"][0], expected, decimal=2)
                        else:
                            solver.update(b=np.array([1.0, 0.0]))
                        seen.append("ok")
                    except ValueError as e:
                        assert "Workspace not initialized" in str(e)
                        seen.append("gone")

            def destroyer():
                barrier.wait(timeout=10)
                raw.__del__()  # what the interpreter runs when the last reference goes (scs_finish under the instance lock)
                raw.__del__()  # idempotent

            _run_threads([lambda: user("solve"), lambda: user("solve"), lambda: user("update"), lambda: user("solve"), destroyer], timeout=30)
            assert len(seen) == 24 and "gone" in seen
            with pytest.raises(ValueError, match="Workspace not initialized"):
                solver.solve()

    def test_solve_after_del_raises_or_succeeds(self, scs):
        """:729-743"""
        data, cone, expected = _make_simple_lp()
        solver = scs.SCS(data, cone, verbose=False)
        sol = solver.solve()
        assert sol["info"]["status_val"] == 1
        assert_almost_equal(sol["x"][0], expected, decimal=2)
        assert solver.solve()["info"]["status_val"] == 1


# ================================================================ R:test/test_free_threading.py:746-874
class TestConcurrentSolveUpdateStress:
    def test_solve_update_barrage_shared_instance(self, scs):
        """:750-796 — 16 threads, 10 rounds each"""
        data, cone, _ = _make_simple_lp()
        solver = scs.SCS(data, cone, verbose=False)

        def solve_worker():
            for _ in range(10):
                assert solver.solve()["info"]["status_val"] in (1, 2)

        def update_worker():
            for _ in range(10):
                solver.update(b=np.array([1.0, 0.0]))

        _run_threads([solve_worker if i % 2 == 0 else update_worker for i in range(16)], timeout=60)
        assert solver.solve()["info"]["status_val"] in (1, 2)

    def test_concurrent_update_different_data(self, scs):
        """:798-835"""
        data, cone, _ = _make_simple_lp()
        solver = scs.SCS(data, cone, verbose=False)
        rng = np.random.RandomState(42)
        b_updates = [np.array([rng.uniform(0.5, 2.0), 0.0]) for _ in range(NUM_THREADS)]
        c_updates = [np.array([rng.uniform(-2.0, -0.5)]) for _ in range(NUM_THREADS)]
        barrier = threading.Barrier(NUM_THREADS)

        def worker(tid):
            barrier.wait(timeout=10)
            for _ in range(5):
                solver.update(b=b_updates[tid], c=c_updates[tid])
                sol = solver.solve()
                assert sol["info"]["status_val"] in (1, 2, -2, -7)
                # whichever thread's update came last, the answer is the optimum of ONE of the 8 x 8 (b, c) combinations: x = b[0]
                assert min(abs(sol["x"][0] - b[0]) for b in b_updates) < 0.05

        _run_threads([lambda t=t: worker(t) for t in range(NUM_THREADS)], timeout=60)

    def test_warm_start_under_contention(self, scs):
        """:837-874"""
        data, cone, expected = _make_simple_lp()
        solver = scs.SCS(data, cone, verbose=False)
        sol0 = solver.solve()
        assert sol0["info"]["status_val"] == 1

        def worker():
            for _ in range(10):
                sol = solver.solve(x=np.array([0.9]), y=sol0["y"], s=sol0["s"])
                assert sol["info"]["status_val"] == 1
                assert_almost_equal(sol["x"][0], expected, decimal=2)

        _run_threads([worker] * NUM_THREADS, timeout=60)


# ================================================================ R:test/test_free_threading.py:877-985
class TestErrorPathContention:
    def test_bad_warm_start_does_not_deadlock(self, scs):
        """:886-937"""
        data, cone, expected = _make_simple_lp()
        solver = scs.SCS(data, cone, verbose=False)
        solver.solve()
        barrier = threading.Barrier(NUM_THREADS)

        def good_worker():
            barrier.wait(timeout=10)
            for _ in range(10):
                sol = solver.solve()
                assert sol["info"]["status_val"] == 1
                assert_almost_equal(sol["x"][0], expected, decimal=2)

        def bad_worker():
            barrier.wait(timeout=10)
            for _ in range(10):
                with pytest.raises(ValueError):
                    solver.solve(x=np.array([1.0, 2.0, 3.0]))  # wrong dimension: the error path must release the lock

        _run_threads([bad_worker] + [good_worker] * (NUM_THREADS - 1), timeout=30)

    def test_bad_update_does_not_deadlock(self, scs):
        """:939-985"""
        data, cone, _ = _make_simple_lp()
        solver = scs.SCS(data, cone, verbose=False)
        barrier = threading.Barrier(NUM_THREADS)

        def good_worker():
            barrier.wait(timeout=10)
            for _ in range(10):
                assert solver.solve()["info"]["status_val"] in (1, 2)

        def bad_worker():
            barrier.wait(timeout=10)
            for _ in range(10):
                with pytest.raises(ValueError):
                    solver.update(b=np.array([1.0, 2.0, 3.0]))

        _run_threads([bad_worker] + [good_worker] * (NUM_THREADS - 1), timeout=30)

    def test_failing_init_next_to_running_solves(self, scs):
        """device backend: `scs_init` throws inside the core (cone dimensions that do not add up: R:test/test_scs_basic.py:99-100;
        an order the dense solver refuses) in some threads while others construct and solve.  The failing thread gets ITS reason
        (scs_hip_last_error is per thread), holds nothing afterwards (stream, pinned block and device blocks go back to their pools),
        and the others never notice."""
        data, cone, expected = _make_simple_lp()
        big_n = 9000
        big = {"A": sp.eye(big_n, format="csc"), "b": np.ones(big_n), "c": np.ones(big_n)}
        barrier = threading.Barrier(NUM_THREADS)

        def good_worker():
            barrier.wait(timeout=10)
            for _ in range(10):
                sol = scs.SCS(data, cone, verbose=False).solve()
                assert sol["info"]["status_val"] == 1
                assert_almost_equal(sol["x"][0], expected, decimal=2)

        def bad_cone_worker():
            barrier.wait(timeout=10)
            for _ in range(10):
                with pytest.raises(ValueError, match=r"ScsWork allocation error! \(cone dimensions do not match m\)"):
                    scs.SCS(data, {"q": [1], "l": 0}, verbose=False)

        def bad_dense_worker():
            barrier.wait(timeout=10)
            for _ in range(5):
                with pytest.raises(ValueError, match=r"hip_dense: n = 9000 exceeds 8192"):
                    scs.SCS(big, {"l": big_n}, verbose=False, linear_solver=scs.LinearSolver.HIP_DENSE)

        _run_threads([bad_cone_worker, bad_dense_worker] + [good_worker] * (NUM_THREADS - 2), timeout=60)


# ================================================================ device-backend scenarios
class TestDeviceBackendConcurrency:
    def test_64_live_workspaces_all_solving(self, scs):
        """more live workspaces than the stream pool creates (32 per device, SCS_HIP_STREAMS): the 33rd .. 64th share streams with
        earlier ones.  All 64 are constructed first, then all solve at once from 16 threads; everyone gets the bits of the solve
        made alone (R:test/test_thread_safety.py:78-93: independent instances, no interference)."""
        K = {"l": 60, "q": [8, 5], "s": [4]}
        proj = lambda z, K: __import__("scs")._scs_hip.proj_cone(z, K, dual=True)  # noqa: E731
        probs = [pg.gen_feasible(K, 30, 6, 500 + i, proj) for i in range(8)]
        solo = [scs.SCS(p[0], K, verbose=False).solve() for p in probs]
        assert all(r["info"]["status_val"] == 1 for r in solo)
        solvers = [scs.SCS(probs[i % 8][0], K, verbose=False) for i in range(64)]
        barrier = threading.Barrier(16)
        results = [None] * 64

        def worker(t):
            barrier.wait(timeout=20)
            for i in range(t, 64, 16):
                results[i] = solvers[i].solve(warm_start=False)

        _run_threads([lambda t=t: worker(t) for t in range(16)], timeout=90)
        for i, r in enumerate(results):
            _same_bits(r, solo[i % 8], "workspace %d" % i)
        # and once more with every workspace warm: shared streams must not mix up the warm-start state
        again = [sv.solve() for sv in solvers[:40:5]]
        for k, r in enumerate(again):
            assert r["info"]["status_val"] == 1 and r["info"]["iter"] <= solo[(5 * k) % 8]["info"]["iter"]

    def test_dense_grouped_and_indirect_paths_from_different_threads(self, scs):
        """the three ways to solve a config-5-shaped problem on the device at once: threads 0-1 the dense direct solver, 2-3 the
        indirect one, 4-5 grouped solves (scs.solve_batch) of four members each — dense members in one, indirect in the other —,
        6-7 PSD-free LPs; every result has the bits of its own solve made alone before the threads start"""
        K = {"l": 80, "q": [10, 6], "s": [5, 5]}
        proj = lambda z, K: __import__("scs")._scs_hip.proj_cone(z, K, dual=True)  # noqa: E731
        probs = [pg.gen_feasible(K, 40, 8, 900 + i, proj)[0] for i in range(4)]
        lp, lp_cone, lp_x = _make_simple_lp()
        DN, IN = scs.LinearSolver.HIP_DENSE, scs.LinearSolver.HIP_INDIRECT
        solo = {ls: [scs.SCS(d, K, verbose=False, linear_solver=ls).solve() for d in probs] for ls in (DN, IN)}
        assert all(r["info"]["status_val"] == 1 for rs in solo.values() for r in rs)
        barrier = threading.Barrier(NUM_THREADS)

        def single(ls):
            barrier.wait(timeout=20)
            for rep in range(3):
                for i, d in enumerate(probs):
                    _same_bits(scs.SCS(d, K, verbose=False, linear_solver=ls).solve(), solo[ls][i], "%s %d" % (ls.value, i))

        def grouped(ls):
            barrier.wait(timeout=20)
            for rep in range(3):
                res = scs.solve_batch([scs.SCS(d, K, verbose=False, linear_solver=ls) for d in probs])
                for i, r in enumerate(res):
                    _same_bits(r, solo[ls][i], "grouped %s %d" % (ls.value, i))

        def lps():
            barrier.wait(timeout=20)
            for rep in range(20):
                assert_almost_equal(scs.SCS(lp, lp_cone, verbose=False).solve()["x"][0], lp_x, decimal=2)

        _run_threads([lambda: single(DN), lambda: single(DN), lambda: single(IN), lambda: single(IN),
                      lambda: grouped(DN), lambda: grouped(IN), lps, lps], timeout=100)

    def test_time_limited_solve_while_another_thread_updates(self, scs):
        """a solve that `time_limit_secs` cuts short (R:scs/scsobject.h:860-868; the status is then the best guess of an unfinished
        run) on a shared instance while another thread keeps calling update(): both serialise on the instance lock, nothing hangs,
        and the instance still solves the updated problem afterwards"""
        data, cone = _make_larger_lp(n=40, seed=7)
        solver = scs.SCS(data, cone, verbose=False, eps_abs=1e-13, eps_rel=1e-13, max_iters=10 ** 6, time_limit_secs=0.02)
        statuses, stop = [], threading.Event()

        def solve_worker():
            for _ in range(6):
                sol = solver.solve(warm_start=False)
                statuses.append((sol["info"]["status_val"], sol["info"]["iter"], sol["info"]["solve_time"]))
                assert np.all(np.isfinite(sol["x"]))
            stop.set()

        def update_worker():
            k = 0
            while not stop.is_set() and k < 10000:
                solver.update(b=data["b"] * (1.0 + 1e-3 * (k % 5)))
                k += 1
            assert k > 0

        _run_threads([solve_worker, update_worker], timeout=60)
        assert len(statuses) == 6
        for st, it, ms in statuses:
            assert it < 10 ** 6 and ms < 2000.0, (st, it, ms)      # cut short by the clock, not by max_iters
            assert st in (1, 2, -6, -7), st                         # solved (before the limit) or the *_INACCURATE guess of a cut-short run
        solver.update(b=data["b"])
        fresh = scs.SCS(data, cone, verbose=False, eps_abs=1e-7, eps_rel=1e-7).solve()
        relaxed = scs.SCS(data, cone, verbose=False, eps_abs=1e-7, eps_rel=1e-7, time_limit_secs=30.0)
        got = relaxed.solve()
        assert got["info"]["status_val"] == 1 and fresh["info"]["status_val"] == 1
        np.testing.assert_allclose(got["x"], fresh["x"], rtol=1e-5, atol=1e-6)
