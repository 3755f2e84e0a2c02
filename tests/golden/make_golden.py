#!/usr/bin/env python3
"""Capture golden vectors from the reference's own pure-Python test tooling.

Runs ONLY in the build container (needs /root/reference); the GPU box never
runs it.  It imports `R:test/gen_random_cone_prob.py` (the only executable
reference arithmetic for this path) and stores NUMBERS
ONLY (inputs and expected outputs) as .npz fixtures next to this script.  The
embedded QP of `R:test/test_warm_start_consistency.py:18-211` is captured the
same way: the module's numeric literals are read with `ast` (nothing from that
file is executed or copied as text).

Usage:  python tests/golden/make_golden.py
"""
import ast
import os
import sys

import numpy as np

sys.dont_write_bytecode = True  # /root/reference is read-only
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "test"))
import gen_random_cone_prob as tools  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def cone_arrays(K):
    """Flatten a cone dict into npz-storable arrays."""
    return {
        "K_z": np.int64(K.get("z", 0)),
        "K_l": np.int64(K.get("l", 0)),
        "K_q": np.asarray(K.get("q", []), dtype=np.int64),
        "K_s": np.asarray(K.get("s", []), dtype=np.int64),
        "K_ep": np.int64(K.get("ep", 0)),
        "K_ed": np.int64(K.get("ed", 0)),
        "K_p": np.asarray(K.get("p", []), dtype=np.float64),
    }


def prob_arrays(prefix, data, p_star=None):
    A = data["A"].tocsc()
    A.sort_indices()
    out = {
        prefix + "A_data": A.data.astype(np.float64),
        prefix + "A_indices": A.indices.astype(np.int32),
        prefix + "A_indptr": A.indptr.astype(np.int32),
        prefix + "shape": np.asarray(A.shape, dtype=np.int64),
        prefix + "b": np.asarray(data["b"], dtype=np.float64),
        prefix + "c": np.asarray(data["c"], dtype=np.float64),
    }
    if p_star is not None:
        out[prefix + "p_star"] = np.float64(p_star)
    return out


# cones used by the reference's certificate tests
K_STD = {  # R:test/test_solve_random_cone_prob.py:33-41
    "z": 10, "l": 15, "q": [5, 10, 0, 1], "s": [3, 4, 0, 0, 1, 10],
    "ep": 10, "ed": 10, "p": [-0.25, 0.5, 0.75, -0.33],
}
K_RAND = {  # R:test/test_scs_rand.py:64-72
    "z": 10, "l": 25, "q": [5, 10, 0, 1], "s": [2, 1, 2, 0, 1],
    "ep": 0, "ed": 0, "p": [0.25, -0.75, 0.33, -0.33, 0.2],
}
K_SDP = {  # R:test/test_scs_sdp.py:64-72
    "z": 10, "l": 25, "q": [5, 10, 0, 1], "s": [2, 1, 2, 0, 1, 10, 8],
    "ep": 0, "ed": 0, "p": [0.25, -0.75, 0.33, -0.33, 0.2],
}


def make_problems():
    out = {}
    # --- standard K: feasible 3000 / infeasible 3001 / unbounded 3002
    m = tools.get_scs_cone_dims(K_STD)
    rng = np.random.RandomState(3000)
    data, p_star = tools.gen_feasible(K_STD, n=m // 3, density=0.1, rng=rng)
    out.update(prob_arrays("std_feas_", data, p_star))
    rng = np.random.RandomState(3001)
    out.update(prob_arrays("std_infeas_", tools.gen_infeasible(K_STD, n=m // 2, rng=rng)))
    rng = np.random.RandomState(3002)
    out.update(prob_arrays("std_unbdd_", tools.gen_unbounded(K_STD, n=m // 2, rng=rng)))
    np.savez_compressed(os.path.join(HERE, "problems_std.npz"), **cone_arrays(K_STD), **out)

    # --- test_scs_rand K: first 3 feasible draws of seed 1000, 2 infeasible (1001), 2 unbounded (1002)
    for name, K, base in (("rand", K_RAND, 1000), ("sdp", K_SDP, 2000)):
        out = {}
        m = tools.get_scs_cone_dims(K)
        rng = np.random.RandomState(base)
        for i in range(3):
            data, p_star = tools.gen_feasible(K, n=m // 3, density=0.1, rng=rng)
            out.update(prob_arrays("feas%d_" % i, data, p_star))
        rng = np.random.RandomState(base + 1)
        for i in range(2):
            out.update(prob_arrays("infeas%d_" % i, tools.gen_infeasible(K, n=m // 2, rng=rng)))
        rng = np.random.RandomState(base + 2)
        for i in range(2):
            out.update(prob_arrays("unbdd%d_" % i, tools.gen_unbounded(K, n=m // 2, rng=rng)))
        np.savez_compressed(os.path.join(HERE, "problems_%s.npz" % name), **cone_arrays(K), **out)

    # --- config-1 LP (BASELINE.json configs[0]): K={l:4000}, n=2000, nnz~1e5, seed 1
    K = {"z": 0, "l": 4000, "q": [], "s": [], "ep": 0, "ed": 0, "p": []}
    rng = np.random.RandomState(1)
    data, p_star = tools.gen_feasible(K, n=2000, density=0.0125, rng=rng)
    out = prob_arrays("lp_", data, p_star)
    # float32 storage of A would change the problem; keep f64 but compressed.
    np.savez_compressed(os.path.join(HERE, "problem_config1_lp.npz"), **cone_arrays(K), **out)


def make_projections():
    """proj_cone / proj_dual_cone goldens per cone type."""
    out = {}
    cases = []
    for q in (1, 2, 5, 10, 100):
        cases.append(("q%d" % q, {"q": [q]}))
    for s in (1, 2, 3, 4, 10, 50):
        cases.append(("s%d" % s, {"s": [s]}))
    cases.append(("ep8", {"ep": 8}))
    cases.append(("ed8", {"ed": 8}))
    for j, p in enumerate((0.5, 0.25, 0.75, -0.25, -0.33)):
        cases.append(("p%d" % j, {"p": [p] * 6}))
    cases.append(("zl", {"z": 7, "l": 9}))
    cases.append(("std", K_STD))
    cases.append(("sdp", K_SDP))
    names = []
    for name, Kpart in cases:
        K = {"z": 0, "l": 0, "q": [], "s": [], "ep": 0, "ed": 0, "p": []}
        K.update(Kpart)
        m = tools.get_scs_cone_dims(K)
        rng = np.random.RandomState(12345 + len(names))
        for k, scl in enumerate((1.0, 10.0, 0.1)):
            z = scl * rng.randn(m)
            tag = "%s_%d_" % (name, k)
            out[tag + "z"] = z
            out[tag + "proj"] = np.asarray(tools.proj_cone(z, K), dtype=np.float64)
            out[tag + "dual"] = np.asarray(tools.proj_dual_cone(z, K), dtype=np.float64)
            for kk, vv in cone_arrays(K).items():
                out[tag + kk] = vv
            names.append(tag)
    out["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(HERE, "cone_projections.npz"), **out)


def make_warm_start_qp():
    """Numeric literals of the embedded QP (R:test/test_warm_start_consistency.py:18-211)."""
    path = os.path.join(REF, "test", "test_warm_start_consistency.py")
    with open(path) as f:
        tree = ast.parse(f.read())
    want = {"_P_data", "_P_indices", "_P_indptr", "_G_data", "_G_indices",
            "_G_indptr", "_q", "_h", "_x0", "_y0", "_s0"}
    vals = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1:
            t = node.targets[0]
            if isinstance(t, ast.Name) and t.id in want:
                v = node.value
                # np.array([...]) or plain list
                if isinstance(v, ast.Call) and v.args:
                    v = v.args[0]
                vals[t.id] = np.asarray(ast.literal_eval(v))
    missing = want - set(vals)
    assert not missing, missing
    np.savez_compressed(
        os.path.join(HERE, "warm_start_qp.npz"),
        P_data=vals["_P_data"].astype(np.float64),
        P_indices=vals["_P_indices"].astype(np.int32),
        P_indptr=vals["_P_indptr"].astype(np.int32),
        G_data=vals["_G_data"].astype(np.float64),
        G_indices=vals["_G_indices"].astype(np.int32),
        G_indptr=vals["_G_indptr"].astype(np.int32),
        q=vals["_q"].astype(np.float64), h=vals["_h"].astype(np.float64),
        x0=vals["_x0"].astype(np.float64), y0=vals["_y0"].astype(np.float64),
        s0=vals["_s0"].astype(np.float64),
    )


if __name__ == "__main__":
    make_problems()
    make_projections()
    make_warm_start_qp()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
