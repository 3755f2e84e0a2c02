"""GPU parity tests of the Anderson accelerator against oracle/oscs_aa.c, step by step.

Both objects expose the interface of the SCS core's aa.c (init / apply / safeguard / reset; named at
R:meson.build:187, knobs R:README.md:98-104, statistics R:scs/scsobject.h:1096-1107).  The driver feeds BOTH the
same (x, F(x)) pairs — the oracle's iterates — so that every single step can be compared: the weights gamma, the
extrapolated iterate, the returned aa_norm, accept / reject, the safeguard verdict with its roll-back, and the
statistics.  Both device formulations are covered: the TSQR path (default) and the incremental Gram path
(SCS_HIP_AA=gram).

Tolerances: gamma and the extrapolated iterate 1e-8 relative on well-conditioned histories (the oracle solves the
Gram system S'Y; the TSQR path forms the same system from the triangle of a Householder QR, so the two agree to
rounding x condition number); decisions (accept / reject / safeguard) and counters must be identical.
"""
import numpy as np
import pytest

import helpers
import problem_gen as pg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from scs import _scs_hip
    assert _scs_hip.device_count() > 0, "GPU tests need a HIP device (no CPU fallback exists)"
    return _scs_hip


@pytest.fixture(scope="module")
def oracle():
    from oracle import scs_oracle
    return scs_oracle


@pytest.fixture(params=["tsqr", "gram"])
def aa_mode(request, monkeypatch):
    monkeypatch.setenv("SCS_HIP_AA", request.param)
    return request.param


def _linear_map(dim, seed, rate=0.9):
    """contraction F(x) = b + d * x + 0.05 * roll(x): spectrum inside (0.05, rate + 0.05)"""
    rng = np.random.RandomState(seed)
    d = rng.uniform(0.0, rate, dim)
    b = rng.randn(dim)
    return (lambda x: b + d * x + 0.05 * np.roll(x, 1)), rng.randn(dim)


def _drive(h, o, F, x0, steps, rtol=1e-8, check_gamma=True, interval=1):
    """run `steps` fixed-point iterations; both accelerators see the ORACLE's iterates"""
    x = x0.copy()
    n_solved = 0
    for k in range(steps):
        f = F(x)
        if k % interval == 0:
            no, fo = o.apply(f, x)
            nh, fh = h.apply(f, x)
            assert (no > 0) == (nh > 0) and (no < 0) == (nh < 0), (k, no, nh)
            scale = max(np.abs(fo).max(), 1.0)
            np.testing.assert_allclose(fh, fo, rtol=0, atol=rtol * scale, err_msg="iterate, step %d" % k)
            if no > 0:
                n_solved += 1
                assert abs(nh - no) <= rtol * 10 * max(no, 1.0), (k, nh, no)
                sh, so = h.stats(), o.stats()   # statistics of THIS solve
                assert sh["last_rank"] == so["last_rank"]
                assert abs(sh["last_regularization"] - so["last_regularization"]) <= 1e-6 * so["last_regularization"], (k, sh, so)
                assert abs(sh["last_aa_norm"] - so["last_aa_norm"]) <= 1e-6 * max(1.0, so["last_aa_norm"])
                if check_gamma:
                    go, gh = o.last_gamma(), h.last_gamma()
                    # (errors of gamma live in the near-null directions of the system matrix — a converging linear map
                    # makes the history nearly dependent — and cancel in f - D gamma: the iterate is the tight check)
                    np.testing.assert_allclose(gh, go, rtol=0, atol=1e-5 * max(np.abs(go).max(), 1.0), err_msg="gamma, step %d" % k)
            x_new = fo
            f_new = F(x_new)
            ro, f2o, x2o = o.safeguard(f_new, x_new)
            rh, f2h, x2h = h.safeguard(f_new, x_new)
            assert ro == rh, (k, ro, rh)
            np.testing.assert_allclose(f2h, f2o, rtol=0, atol=rtol * scale)
            np.testing.assert_allclose(x2h, x2o, rtol=0, atol=rtol * scale)
            x = f2o
        else:
            x = f
    return n_solved


def _same_counters(h, o, exact_categories=True):
    sh, so = h.stats(), o.stats()
    keys = ["iter", "n_accept", "n_safeguard_reject"]
    if exact_categories:
        keys += ["n_reject_lapack", "n_reject_rank0", "n_reject_nonfinite", "n_reject_weight_cap", "last_rank"]
    for k in keys:
        assert sh[k] == so[k], (k, sh, so)
    return sh, so


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
@pytest.mark.parametrize("dim,mem", [(50, 10), (5000, 10), (5000, 3), (400000, 10), (70000, 19), (3000, 32), (777, 1)])
def test_aa_steps_match_oracle(hip, oracle, aa_mode, type1, dim, mem):
    F, x0 = _linear_map(dim, 3 + dim % 7)
    h = hip.AndersonAccelerator(dim, mem, type1=type1)
    o = oracle.OracleAA(dim, mem, type1=type1)
    steps = 3 * mem + 12
    # a long history of a LINEAR map becomes numerically dependent as the iteration converges: compare gamma only
    # while the Gram matrix is well conditioned (short histories), the iterate always
    n = _drive(h, o, F, x0, steps, rtol=1e-7 if mem >= 19 else 1e-8, check_gamma=mem <= 10)
    assert n > 0
    sh, so = _same_counters(h, o)
    assert sh["n_accept"] > 0


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
@pytest.mark.parametrize("dim,mem", [(30000, 50), (9000, 80), (2000, 33)])
def test_aa_lookback_beyond_the_register_block(hip, oracle, type1, dim, mem):
    """acceleration_lookback > 32 (the reference accepts any, R:scs/scsobject.h:545): histories longer than one register
    block take the Gram path a block of 32 columns per launch — step-by-step parity with the oracle as above.  The map
    has a wide spectrum so that a 50 - 80 column history stays numerically independent for a while."""
    rng = np.random.RandomState(mem)
    d = rng.uniform(-0.95, 0.95, dim)
    b = rng.randn(dim)
    F = lambda x: b + d * x + 0.04 * np.roll(x, 1)  # noqa: E731
    h = hip.AndersonAccelerator(dim, mem, type1=type1, regularization=1e-6)
    o = oracle.OracleAA(dim, mem, type1=type1, regularization=1e-6)
    n = _drive(h, o, F, rng.randn(dim), mem + 25, rtol=1e-6, check_gamma=False)
    assert n > 0
    sh, so = _same_counters(h, o)
    assert sh["iter"] == mem + 25


def test_full_solve_with_lookback_50_matches_oracle(hip, oracle):
    """a whole solve with acceleration_lookback = 50 (was refused by this backend in rounds 1-2)"""
    K, n, k, seed = pg.workload("small_lp_soc")
    data, p_star, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: oracle.proj_cone(z, K, dual=True))
    args = helpers.raw_args(data, K)
    stg = dict(STG, acceleration_lookback=50, acceleration_interval=1, eps_abs=1e-7, eps_rel=1e-7)
    got = hip.SCS(*args, **stg).solve(False, None, None, None)
    ref = oracle.OracleSCS(*args, indirect=False, **stg).solve(False)
    assert got["info"]["status"] == ref["info"]["status"] == "solved"
    ga, ra = got["info"]["aa_stats"], ref["info"]["aa_stats"]
    solves = lambda a: a["n_accept"] + a["n_reject_lapack"] + a["n_reject_rank0"] + a["n_reject_nonfinite"] + a["n_reject_weight_cap"]  # noqa: E731
    assert ga["iter"] > 50 and solves(ga) > 0 and solves(ra) > 0, (ga, ra)  # the 50-column system was really solved
    assert abs(got["info"]["pobj"] - p_star) < 1e-6 * max(1, abs(p_star))
    for key in ("x", "y", "s"):
        np.testing.assert_allclose(got[key], ref[key], rtol=1e-4, atol=1e-4 * np.abs(ref[key]).max(), err_msg=key)


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
@pytest.mark.parametrize("relaxation,regularization", [(0.8, 1e-8), (1.0, 0.0), (1.3, 1e-4)])
def test_aa_relaxation_and_regularization(hip, oracle, aa_mode, type1, relaxation, regularization):
    dim, mem = 20000, 5
    F, x0 = _linear_map(dim, 11, rate=0.95)
    kw = dict(type1=type1, relaxation=relaxation, regularization=regularization)
    h, o = hip.AndersonAccelerator(dim, mem, **kw), oracle.OracleAA(dim, mem, **kw)
    assert _drive(h, o, F, x0, 30) > 0
    _same_counters(h, o)


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
def test_aa_interval_and_reset(hip, oracle, aa_mode, type1):
    """acceleration_interval = 5 (plain steps in between) and an external reset (what a scale update does)"""
    dim, mem = 9000, 4
    F, x0 = _linear_map(dim, 5)
    h, o = hip.AndersonAccelerator(dim, mem, type1=type1), oracle.OracleAA(dim, mem, type1=type1)
    _drive(h, o, F, x0, 60, interval=5)
    h.reset()
    o.reset()
    assert _drive(h, o, F, F(x0), 40, interval=2) > 0
    _same_counters(h, o)


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
def test_aa_weight_cap_rejects(hip, oracle, aa_mode, type1):
    """||gamma|| >= max_weight_norm: the step is refused, the iterate untouched, the history restarted"""
    dim, mem = 6000, 4
    F, x0 = _linear_map(dim, 21)
    kw = dict(type1=type1, max_weight_norm=0.05)  # every real solve exceeds it
    h, o = hip.AndersonAccelerator(dim, mem, **kw), oracle.OracleAA(dim, mem, **kw)
    _drive(h, o, F, x0, 25)
    sh, so = _same_counters(h, o)
    assert sh["n_reject_weight_cap"] > 0 and sh["n_accept"] == 0


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
def test_aa_rank_deficient_history_regularized(hip, oracle, aa_mode, type1):
    """A map with one-dimensional dynamics: every column of S (and Y) is a multiple of the same vector, the system
    matrix has rank one.  With the default regularisation both solve it, and the extrapolated iterates agree (gamma
    itself is not unique)."""
    dim, mem = 4000, 4
    rng = np.random.RandomState(2)
    xs, dvec = rng.randn(dim), rng.randn(dim)
    F = lambda x: xs + 0.5 * (dvec @ (x - xs)) / (dvec @ dvec) * dvec   # noqa: E731
    h, o = hip.AndersonAccelerator(dim, mem, type1=type1), oracle.OracleAA(dim, mem, type1=type1)
    x = xs + 3.0 * dvec
    for k in range(12):
        f = F(x)
        no, fo = o.apply(f, x)
        nh, fh = h.apply(f, x)
        assert (no > 0) == (nh > 0) and (no < 0) == (nh < 0), (k, no, nh)
        np.testing.assert_allclose(fh, fo, rtol=0, atol=1e-6 * np.abs(fo).max())
        ro, f2o, x2o = o.safeguard(F(fo), fo)
        rh, f2h, x2h = h.safeguard(F(fo), fo)
        assert ro == rh
        x = f2o
    _same_counters(h, o, exact_categories=False)


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
def test_aa_exactly_singular_history_is_refused(hip, oracle, aa_mode, type1):
    """regularisation 0 and an EXACTLY rank-one history: x_k = 2^-k d with an integer vector d and F(x) = x / 2, so
    every entry of S, Y and every dot product is exact in binary floating point and the system matrix is singular
    to the last bit.  Every solve must be refused and f left alone.  (Which counter fires may differ: the oracle's LU
    meets an exactly zero pivot; the triangle of the Householder QR carries rounding, so the TSQR path may meet a
    1e-30 pivot and trip the weight cap instead.)"""
    dim, mem = 3000, 3
    d = np.random.RandomState(5).randint(-8, 9, dim).astype(np.float64)
    h = hip.AndersonAccelerator(dim, mem, type1=type1, regularization=0.0)
    o = oracle.OracleAA(dim, mem, type1=type1, regularization=0.0)
    solves = 0
    for k in range(10):
        x = d * 2.0 ** (-k)
        f = 0.5 * x
        no, fo = o.apply(f, x)
        nh, fh = h.apply(f, x)
        assert no <= 0 and nh <= 0, (k, no, nh)
        assert (no < 0) == (nh < 0), (k, no, nh)
        solves += no < 0
        np.testing.assert_array_equal(fo, f)
        np.testing.assert_array_equal(fh, f)
    assert solves >= 2
    sh, so = _same_counters(h, o, exact_categories=False)
    rej = lambda s: s["n_reject_lapack"] + s["n_reject_rank0"] + s["n_reject_nonfinite"] + s["n_reject_weight_cap"]  # noqa: E731
    assert rej(sh) == rej(so) == solves and sh["n_accept"] == 0


def test_aa_zero_history_is_rank0(hip, oracle, aa_mode):
    """x == F(x) from the start: S = Y = 0, the system matrix is exactly zero => rank 0 in both"""
    dim, mem = 1000, 3
    h, o = hip.AndersonAccelerator(dim, mem), oracle.OracleAA(dim, mem)
    x = np.random.RandomState(0).randn(dim)
    for k in range(8):
        no, fo = o.apply(x, x)
        nh, fh = h.apply(x, x)
        assert (no < 0) == (nh < 0)
        np.testing.assert_array_equal(fh, x)
    sh, so = _same_counters(h, o)
    assert sh["n_reject_rank0"] > 0 and sh["last_rank"] == 0


def test_aa_nonfinite_input_rejected(hip, oracle, aa_mode):
    dim, mem = 3000, 3
    F, x0 = _linear_map(dim, 8)
    h, o = hip.AndersonAccelerator(dim, mem), oracle.OracleAA(dim, mem)
    x = x0
    for k in range(10):
        f = F(x)
        if k == 6:
            f = f.copy()
            f[17] = np.nan
        no, fo = o.apply(f, x)
        nh, fh = h.apply(f, x)
        assert (no > 0) == (nh > 0) and (no < 0) == (nh < 0), (k, no, nh)
        if k == 6:
            assert no < 0
        x = np.where(np.isfinite(fo), fo, 0.0)
    sh, so = _same_counters(h, o, exact_categories=False)


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
def test_aa_safeguard_rolls_back(hip, oracle, aa_mode, type1):
    """the step after an accepted extrapolation has a LARGER fixed-point residual: both restore the pre-extrapolation
    pair (x, F(x)) bit for bit and restart the history"""
    dim, mem = 5000, 3
    F, x0 = _linear_map(dim, 13)
    h, o = hip.AndersonAccelerator(dim, mem, type1=type1), oracle.OracleAA(dim, mem, type1=type1)
    x = x0
    fired = False
    for k in range(14):
        f = F(x)
        no, fo = o.apply(f, x)
        nh, fh = h.apply(f, x)
        assert (no > 0) == (nh > 0)
        x_new = fo
        f_new = F(x_new)
        if no > 0 and k >= 6 and not fired:   # sabotage: pretend the map threw the iterate far away
            f_new = x_new + 1e3
            ro, f2o, x2o = o.safeguard(f_new, x_new)
            rh, f2h, x2h = h.safeguard(f_new, x_new)
            assert ro == rh == -1
            np.testing.assert_array_equal(f2o, f)
            np.testing.assert_array_equal(x2o, x)
            np.testing.assert_array_equal(f2h, f)   # restored exactly: they are copies
            np.testing.assert_array_equal(x2h, x)
            fired = True
        else:
            ro, f2o, x2o = o.safeguard(f_new, x_new)
            rh, f2h, x2h = h.safeguard(f_new, x_new)
            assert ro == rh
        x = f2o
    assert fired
    sh, so = _same_counters(h, o)
    assert sh["n_safeguard_reject"] == 1


def test_aa_deterministic_bits(hip, aa_mode):
    """fixed-order trees everywhere: two runs of the same sequence give the same bits"""
    dim, mem = 123457, 10
    F, x0 = _linear_map(dim, 4)
    outs = []
    for rep in range(2):
        h = hip.AndersonAccelerator(dim, mem)
        x = x0
        for k in range(16):
            _, x = h.apply(F(x), x)
        outs.append(x)
    np.testing.assert_array_equal(outs[0], outs[1])


STG = dict(eps_abs=1e-9, eps_rel=1e-9, eps_infeas=1e-9, verbose=False)


def _accel_cases(oracle):
    proj = lambda z, K: oracle.proj_cone(z, K, dual=True)  # noqa: E731
    K, n, k, seed = pg.workload("small_lp_soc")
    d1, p1, _ = pg.gen_feasible(K, n, k, seed, proj)
    K2 = {"z": 10, "l": 600, "q": [30, 12, 5], "s": [6, 3], "ep": 4, "p": [0.4, -0.7]}
    d2, p2, _ = pg.gen_feasible_qp(K2, 400, 7, 5, proj)
    return {"lp_soc": (d1, K, p1), "qp_mixed": (d2, K2, p2)}


@pytest.mark.parametrize("type1", [True, False], ids=["type-I", "type-II"])
@pytest.mark.parametrize("case,interval", [("lp_soc", 10), ("lp_soc", 1), ("qp_mixed", 10)])
def test_full_solve_acceleration_counts_track_oracle(hip, oracle, aa_mode, type1, case, interval):
    """Whole ADMM solves with acceleration ON (1300 - 1600 iterations), adaptive scaling off (its threshold decisions
    amplify last-bit differences): Anderson steps are really taken (n_accept > 0), and the HIP path's iteration count
    and accept / reject counters follow the oracle's CPU-CG variant — the same algorithm with the same linear-solve
    tolerances (observed: identical counts).  interval = 1 exercises the safeguard verdict being consumed before the
    next history update."""
    data, K, p_star = _accel_cases(oracle)[case]
    args = helpers.raw_args(data, K)
    stg = dict(STG, adaptive_scale=False, acceleration_lookback=10, acceleration_interval=interval,
               acceleration_type_1=type1, eps_abs=1e-5, eps_rel=1e-5, max_iters=20000)
    got = hip.SCS(*args, **stg).solve(False, None, None, None)
    ref = oracle.OracleSCS(*args, indirect=True, **stg).solve(False)
    gi, ri = got["info"], ref["info"]
    assert gi["status"] == ri["status"] == "solved", (gi, ri)
    assert abs(gi["pobj"] - p_star) < 1e-3 * max(1, abs(p_star))
    assert ri["iter"] >= 300 and gi["iter"] >= 300
    ga, ra = gi["aa_stats"], ri["aa_stats"]
    assert ga["n_accept"] > 0 and ra["n_accept"] > 0
    # (observed: identical counts for lp_soc.  The mixed-cone QP is chaotic under acceleration: Anderson steps amplify
    # last-bit differences of the reductions.  Seed sweep, profiles/r03_aa_drift_sweep.txt (tools/dbg/aa_drift_sweep.py,
    # seeds 5..12, both types): HIP vs the oracle's CG variant -28 % .. +34 % (one outlier +80 %), median |drift| 10 % —
    # and the oracle against ITSELF, LDL' vs CG linear solves on the same instances, -57 % .. +23 %.  The margin below is
    # what this instance (seed 5: +11 % / -10 %) needs with room for a different reduction order, not a claim about
    # the method.)
    tol = 0.1 if case == "lp_soc" else 0.4
    assert abs(gi["iter"] - ri["iter"]) <= tol * ri["iter"] + 25, (gi["iter"], ri["iter"])
    assert abs(ga["n_accept"] - ra["n_accept"]) <= tol * ra["iter"] + 2, (ga, ra)
    assert abs(gi["rejected_accel_steps"] - ri["rejected_accel_steps"]) <= tol * ra["iter"] + 2, (gi, ri)
    # every Anderson call is followed by its safeguard, except the one of the iteration that met the stopping rule
    assert ga["iter"] - 1 <= gi["accepted_accel_steps"] + gi["rejected_accel_steps"] <= ga["iter"]
    assert ra["iter"] - 1 <= ri["accepted_accel_steps"] + ri["rejected_accel_steps"] <= ra["iter"]
    for key in ("x", "s"):
        np.testing.assert_allclose(got[key], ref[key], rtol=0, atol=1e-3 * np.abs(ref[key]).max(), err_msg=key)


def test_full_solve_first_acceleration_steps_identical(hip, oracle, aa_mode):
    """Before chaos sets in: with max_iters just past the first extrapolations the two runs have made the same
    decisions (same number of calls, accepts, safeguard rejections) and their iterates agree to 1e-6."""
    K, n, k, seed = pg.workload("small_lp_soc")
    data, p_star, _ = pg.gen_feasible(K, n, k, seed, lambda z, K: oracle.proj_cone(z, K, dual=True))
    args = helpers.raw_args(data, K)
    stg = dict(STG, adaptive_scale=False, acceleration_lookback=5, acceleration_interval=5, max_iters=61, eps_abs=0, eps_rel=0)
    got = hip.SCS(*args, **stg).solve(False, None, None, None)
    ref = oracle.OracleSCS(*args, indirect=True, **stg).solve(False)
    ga, ra = got["info"]["aa_stats"], ref["info"]["aa_stats"]
    assert ga["iter"] == ra["iter"] == 12
    assert ga["n_accept"] == ra["n_accept"] and ga["n_accept"] > 0
    assert ga["n_safeguard_reject"] == ra["n_safeguard_reject"]
    assert got["info"]["accepted_accel_steps"] == ref["info"]["accepted_accel_steps"]
    for key in ("x", "y", "s"):
        np.testing.assert_allclose(got[key], ref[key], rtol=0, atol=1e-5 * np.abs(ref[key]).max(), err_msg=key)
