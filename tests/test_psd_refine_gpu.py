"""K9 round 5: the GEMM-only refinement stage of the split pipeline (csrc/psd.hpp psd_stop_test) against LAPACK.

The reference projects PSD cones with LAPACK's syev under USE_LAPACK (R:meson.build:145-147,188; spec
R:test/gen_random_cone_prob.py:153-173): numpy's eigh is that oracle here.  What is tested is the WARM path — sequences of
slowly moving matrices through one workspace (scs_hip_proj_cone_seq), as inside the ADMM loop — because that is where the
refinement replaces the last Jacobi sweeps."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from scs import _scs_hip
    assert _scs_hip.device_count() > 0, "GPU tests need a HIP device (no CPU fallback exists)"
    return _scs_hip


def _lapack(z, orders, o=0):
    out = np.array(z, copy=True)
    for k in orders:
        d = k * (k + 1) // 2
        w, U = np.linalg.eigh(helpers.svec_to_sym(z[o:o + d], k))
        out[o:o + d] = helpers.sym_to_svec((U * np.maximum(w, 0.0)) @ U.T)
        o += d
    return out


def _moving_sequence(rng, orders, steps, spectrum):
    """matrices with a prescribed spectrum whose entries move by `step` (relative) from call to call"""
    base = []
    for k in orders:
        Q, _ = np.linalg.qr(rng.randn(k, k))
        base.append((Q * spectrum(k)) @ Q.T)
    seq = []
    cur = [b.copy() for b in base]
    for st in steps:
        for i, k in enumerate(orders):
            E = rng.randn(k, k)
            cur[i] = cur[i] + st * np.linalg.norm(cur[i]) / k * (E + E.T) / 2
        seq.append(np.concatenate([helpers.sym_to_svec(M) for M in cur]))
    return np.array(seq)


SPECTRA = {
    # both signs, eigenvalues spread over a decade and a half, a gap around zero (what config 4 projects late in a solve)
    "gap": lambda k: np.r_[np.linspace(0.05, 3.0, k // 2), -np.linspace(0.08, 2.0, k - k // 2)],
    # rank-deficient on both sides: clusters of equal eigenvalues inside the positive and the negative block
    "clusters": lambda k: np.r_[np.ones(k // 3), 2.5 * np.ones(k // 3), -1.5 * np.ones(k - 2 * (k // 3))],
    # eigenvalues that come arbitrarily close to zero from both sides: no refinement may be attempted across that pair
    "near_zero": lambda k: np.r_[np.geomspace(1e-9, 1.0, k // 2), -np.geomspace(1e-8, 2.0, k - k // 2)],
}


@pytest.mark.parametrize("spectrum", sorted(SPECTRA))
@pytest.mark.parametrize("mc", ["4", "1"])
def test_refined_projections_of_a_moving_sequence_vs_lapack(hip, monkeypatch, spectrum, mc):
    monkeypatch.setenv("SCS_HIP_PSD_SPLIT", "1")
    monkeypatch.setenv("SCS_HIP_PSD_MC", mc)  # 4: sweeps of one matrix over four CUs; 1: the one-workgroup sweep kernel
    rng = np.random.RandomState(17)
    orders = [200, 96, 130, 40]
    steps = [0.0, 1e-2, 1e-3, 3e-4, 1e-4, 1e-4, 1e-4, 3e-5, 1e-5, 1e-5, 1e-6, 1e-6, 1e-7, 0.0, 1e-3, 1e-5]
    zs = _moving_sequence(rng, orders, steps, SPECTRA[spectrum])
    K = {"s": orders}
    got, stats = hip.proj_cone_seq(zs, K, stats_cap=len(orders))
    for c in range(len(steps)):
        want = _lapack(zs[c], orders)
        o = 0
        for k in orders:
            d = k * (k + 1) // 2
            scale = np.abs(zs[c][o:o + d]).max()
            np.testing.assert_allclose(got[c][o:o + d], want[o:o + d], rtol=0, atol=2e-10 * k * scale,
                                       err_msg="call %d order %d" % (c, k))
            o += d
    assert stats.shape == (len(orders), 8)
    if spectrum == "gap":  # the stage really ran, and what it left passed its own test
        assert (stats[:, 0] >= 5).all(), stats
        assert (stats[:, 1] <= 1).all(), stats
    if spectrum == "near_zero":  # whatever path each call took, it is still the projection (asserted above)
        assert (stats[:, 1] <= stats[:, 0]).all()


def test_refinement_is_deterministic_and_independent_of_the_group_size(hip, monkeypatch):
    """same gate decisions and the same GEMM sequences whatever spreads the sweeps: the bits of G = 1 (one-workgroup sweep
    kernel), 2, 4 agree, and two runs agree with each other"""
    monkeypatch.setenv("SCS_HIP_PSD_SPLIT", "1")
    rng = np.random.RandomState(3)
    orders = [200, 64, 150]
    zs = _moving_sequence(rng, orders, [0.0, 1e-3, 1e-4, 1e-5, 1e-4, 1e-6], SPECTRA["gap"])
    out = {}
    for G in ("1", "2", "4", "4b"):
        monkeypatch.setenv("SCS_HIP_PSD_MC", G[0])
        out[G], st = hip.proj_cone_seq(zs, {"s": orders}, stats_cap=3)
        assert (st[:, 0] >= 3).all(), st
    for G in ("2", "4", "4b"):
        np.testing.assert_array_equal(out["1"], out[G], err_msg="G=" + G)


def test_refinement_switched_off_restores_the_strict_sweeps(hip, monkeypatch):
    """SCS_HIP_PSD_REFINE=0: the split pipeline is bit-identical to the one-launch kernel again (rounds 1-4)"""
    rng = np.random.RandomState(5)
    orders = [100, 40, 64]
    zs = _moving_sequence(rng, orders, [0.0, 1e-3, 1e-5, 1e-5], SPECTRA["gap"])
    monkeypatch.setenv("SCS_HIP_PSD_REFINE", "0")
    out = {}
    for split in ("0", "1"):
        monkeypatch.setenv("SCS_HIP_PSD_SPLIT", split)
        out[split], st = hip.proj_cone_seq(zs, {"s": orders}, stats_cap=3)
        assert (st[:, 0] == 0).all()
    np.testing.assert_array_equal(out["0"], out["1"])
    monkeypatch.setenv("SCS_HIP_PSD_REFINE", "1")
    monkeypatch.setenv("SCS_HIP_PSD_SPLIT", "1")
    ref, st = hip.proj_cone_seq(zs, {"s": orders}, stats_cap=3)
    assert (st[:, 0] >= 2).all(), st
    np.testing.assert_allclose(ref, out["1"], rtol=0, atol=1e-9 * np.abs(zs).max())
