"""CPU tests (gloo) of the N>1 path at world size 2 and at the node's 8: problems sharded round-robin over ranks, no
data-path collective, one gather to rank 0 (scs/batch.py).  The solver is injected (the oracle)
because the product backend needs a GPU; the distributed logic under test is identical.  The batches are ragged on purpose
(different sizes; 11 problems over 8 ranks = shards of 2 and 1; 5 over 8 = three ranks with nothing but the gather)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "scs-python_amd")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch.distributed as dist
    from oracle import scs_oracle
    import problem_gen as pg
    from scs import batch   # importing the package loads libscs_hip.so; no device is touched
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    proj = lambda z, K: scs_oracle.proj_cone(z, K, dual=True)
    problems = []
    NPROB = int(os.environ["SCS_TEST_NPROB"])
    for i in range(NPROB):   # ragged on purpose: different sizes, a count the world size does not divide
        K = {"l": 30 + 7 * i, "q": [4, 3 + i]}
        data, p_star, _ = pg.gen_feasible(K, 20 + 3 * i, 6, 100 + i, proj)
        problems.append((data, K, dict(eps_abs=1e-5, eps_rel=1e-5, verbose=False)))
    calls = []
    def solve_fn(data, cone, settings):
        calls.append(len(data["c"]))
        return scs_oracle.solve(data, cone, indirect=True, **settings)
    res = batch.solve_sharded(problems, solve_fn=solve_fn)
    assert len(calls) == len(batch.shard_indices(NPROB, rank, world)) and calls == [len(problems[i][0]["c"]) for i in batch.shard_indices(NPROB, rank, world)]
    if rank == 0:
        ok = True
        for i, (data, K, st) in enumerate(problems):
            ref = scs_oracle.solve(data, K, indirect=True, **st)
            r = res[i]
            ok &= r["info"]["status_val"] == ref["info"]["status_val"] == 1
            ok &= r["info"]["iter"] == ref["info"]["iter"]
            ok &= bool(np.allclose(r["x"], ref["x"], atol=0, rtol=0)) and bool(np.allclose(r["y"], ref["y"], atol=0, rtol=0))
            ok &= r["x"].size == len(data["c"]) and r["s"].size == len(data["b"])
        print("RESULT", json.dumps({"ok": bool(ok), "n": len(res)}))
    else:
        assert res is None
    dist.destroy_process_group()
''')


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,nprob", [(2, 5), (8, 11), (8, 5)])
def test_sharded_batch_gloo(tmp_path, world, nprob):
    """(8, 11): the shape of the driver's 8-GPU run in miniature — ragged shards of 2 and 1; (8, 5): ranks 5-7 own no problem and
    still take part in the one gather (BASELINE config 5)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1", SCS_TEST_NPROB=str(nprob))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line and '"ok": true' in line[0] and '"n": %d' % nprob in line[0], out.stdout[-2000:]


def test_shard_indices_cover_everything():
    from scs import batch
    for n in (0, 1, 5, 512):
        for w in (1, 2, 3, 8):
            got = sorted(i for r in range(w) for i in batch.shard_indices(n, r, w))
            assert got == list(range(n))


def test_pack_unpack_roundtrip():
    from scs import batch
    rng = np.random.RandomState(0)
    sol = {"x": rng.randn(7), "y": rng.randn(11), "s": rng.randn(11),
           "info": {"status_val": 1, "iter": 125, "pobj": 1.5, "dobj": 1.25, "solve_time": 3.5, "cg_iters": 77}}
    back = batch.unpack_result(batch.pack_result(sol, 8 + 7 + 22 + 13))
    for k in "xys":
        np.testing.assert_array_equal(back[k], sol[k])
    assert back["info"]["iter"] == 125 and back["info"]["cg_iters"] == 77
