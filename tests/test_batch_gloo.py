"""world_size-2 CPU test (gloo) of the N>1 path: problems sharded round-robin over ranks, no
data-path collective, one gather to rank 0 (scs/batch.py).  The solver is injected (the oracle)
because the product backend needs a GPU; the distributed logic under test is identical."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

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
    for i in range(5):   # ragged on purpose: 5 problems over 2 ranks, different sizes
        K = {"l": 30 + 7 * i, "q": [4, 3 + i]}
        data, p_star, _ = pg.gen_feasible(K, 20 + 3 * i, 6, 100 + i, proj)
        problems.append((data, K, dict(eps_abs=1e-5, eps_rel=1e-5, verbose=False)))
    calls = []
    def solve_fn(data, cone, settings):
        calls.append(len(data["c"]))
        return scs_oracle.solve(data, cone, indirect=True, **settings)
    res = batch.solve_sharded(problems, solve_fn=solve_fn)
    assert len(calls) == len(batch.shard_indices(5, rank, world))
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


def test_sharded_batch_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line and '"ok": true' in line[0] and '"n": 5' in line[0], out.stdout[-2000:]


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
