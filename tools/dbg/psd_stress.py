#!/usr/bin/env python3
"""One-off stress of the K9 split pipeline (multi-CU sweeps, look-ahead): random batches of random orders through proj_cone, each checked
against numpy's eigensolver; then repeated calls on a solver instance (warm path).  GPU box: python tools/dbg/psd_stress.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scs-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers
from scs import _scs_hip as hip

rng = np.random.RandomState(2024)
t0 = time.time()
worst = 0.0
for trial in range(40):
    count = int(rng.choice([1, 2, 3, 5, 8, 9, 17, 40, 64, 100, 128]))
    top = int(rng.choice([40, 64, 100, 128, 200, 300]))
    if count * top * top > 6e6:
        count = max(1, int(6e6 / (top * top)))
    orders = [int(rng.randint(33, top + 1)) for _ in range(count)]
    if trial % 5 == 0:
        orders += [int(rng.randint(1, 33)) for _ in range(5)]  # some one-wavefront matrices in the same cone list
    mats = []
    for k in orders:
        M = rng.randn(k, k); M = (M + M.T) / 2
        if rng.rand() < 0.3:
            w, V = np.linalg.eigh(M); w[rng.rand(k) < 0.5] = 0.0; M = (V * w) @ V.T  # rank-deficient
        mats.append(M * 10.0 ** rng.uniform(-3, 3))
    K = {"s": orders}
    z = np.concatenate([helpers.sym_to_svec(M) for M in mats])
    got = hip.proj_cone(z, K)
    o = 0
    for M in mats:
        k = M.shape[0]; d = k * (k + 1) // 2
        w, V = np.linalg.eigh(M)
        want = helpers.sym_to_svec((V * np.maximum(w, 0)) @ V.T)
        err = np.abs(got[o:o + d] - want).max() / max(np.abs(M).max(), 1e-300)
        worst = max(worst, err / k)
        assert err < 2e-11 * k, (trial, k, err)
        o += d
    print("trial %2d: %3d matrices, orders %d..%d ok (%.1f s)" % (trial, len(orders), min(orders), max(orders), time.time() - t0), flush=True)
print("worst error / order: %.2e" % worst)
