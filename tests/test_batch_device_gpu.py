"""GPU tests of the device-resident result path of scs/batch.py: the solver's final (x, y, s) leave its
HBM buffers device-to-device (scs_hip_solution_to_device) into the payload tensor the RCCL gather reads — same bits
as the host copies, NaN vectors for certificates included."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_solution_to_device_matches_host_copies():
    import torch
    import scs
    for fname, prefix in (("problems_std.npz", "std_feas_"), ("problems_std.npz", "std_infeas_"), ("problems_std.npz", "std_unbdd_")):
        data, K, _ = helpers.load_problem(fname, prefix)
        solver = scs.SCS(data, K, verbose=False)
        with pytest.raises(RuntimeError, match="no solution yet"):
            solver._solver.solution_to_device(1, None, None)
        sol = solver.solve()
        m, n = data["A"].shape
        buf = torch.full((n + 2 * m,), 7.0, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        p = buf.data_ptr()
        solver._solver.solution_to_device(p, p + 8 * n, p + 8 * (n + m))
        h = buf.cpu().numpy()
        for key, sl in (("x", slice(0, n)), ("y", slice(n, n + m)), ("s", slice(n + m, n + 2 * m))):
            np.testing.assert_array_equal(h[sl], sol[key], err_msg="%s %s" % (prefix, key))  # NaN == NaN position-wise
        solver._solver.solution_to_device(None, p, None)  # any subset
        np.testing.assert_array_equal(buf.cpu().numpy()[:m], sol["y"])


WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "scs-python_amd")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch
    import torch.distributed as dist
    import scs
    from scs import batch, _scs_hip
    import helpers
    torch.cuda.set_device(0)
    _scs_hip.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    probs = []
    for fname, prefix in (("problems_std.npz", "std_feas_"), ("problems_rand.npz", "feas0_"), ("problems_std.npz", "std_infeas_"),
                          ("problems_sdp.npz", "feas1_"), ("problems_rand.npz", "feas1_"), ("problems_std.npz", "std_unbdd_")):
        data, K, _ = helpers.load_problem(fname, prefix)
        probs.append((data, K, dict(verbose=False)))
    res = batch.solve_sharded(probs, threads=3)
    ok = True
    for (data, K, st), r in zip(probs, res):
        ref = scs.SCS(data, K, linear_solver=scs.LinearSolver.HIP_INDIRECT, **st).solve()
        ok &= r["info"]["status_val"] == ref["info"]["status_val"] and r["info"]["iter"] == ref["info"]["iter"]
        for key in ("x", "y", "s"):
            ok &= bool(np.array_equal(r[key], ref[key], equal_nan=True))
    print("RESULT", json.dumps({"ok": bool(ok), "n": len(res), "backend": dist.get_backend()}))
    dist.destroy_process_group()
''')


def test_sharded_batch_rccl_device_payload(tmp_path):
    """one rank, backend nccl (= RCCL): the payload tensor is filled device to device by 3 concurrent solver threads and
    gathered; results equal direct solves bit for bit (feasible, infeasible and unbounded instances)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out == {"ok": True, "n": 6, "backend": "nccl"}
