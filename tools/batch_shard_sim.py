"""What an N-GPU node would measure on the config-5 batch, simulated on ONE GPU: the 512 problems are sharded round-robin over N ranks
as scs/batch.py does (problem i -> rank i mod N), every shard is solved here as one grouped solve, one after the other, and the job's
wall clock is the SLOWEST shard's (the ranks run concurrently on a node; the single gather is ~5 MB per rank).  python tools/batch_shard_sim.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import numpy as np
import torch
import scs
from scs import _scs_hip, batch as scs_batch
import problem_gen as pg

NB = 512
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
Kb, nb_, kb_, seedb = pg.workload("config5_small")
extra = {"linear_solver": os.environ["LINSYS"]} if os.environ.get("LINSYS") else {}   # hip_dense / hip_indirect
problems = [(pg.gen_feasible(Kb, nb_, kb_, seedb + i, proj)[0], Kb, dict(verbose=False, **extra)) for i in range(NB)]
scs.SCS(problems[0][0], Kb, verbose=False, max_iters=50, **extra).solve()
torch.cuda.synchronize()
for world in (1, 2, 4, 8):
    walls, iters, worst = [], 0, 0
    for rank in range(world):
        shard = [problems[i] for i in scs_batch.shard_indices(NB, rank, world)]
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = scs_batch.solve_sharded(shard, threads=16, grouped=True)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t)
        iters += sum(r["info"]["iter"] for r in res)
        worst = max(worst, max(r["info"]["iter"] for r in res))
        assert all(r["info"]["status_val"] == 1 for r in res)
    print(json.dumps({"n_gpus_simulated": world, "slowest_shard_s": round(max(walls), 3), "shards_s": [round(w, 2) for w in walls],
                      "total_iters": iters, "longest_solve_iters": worst, "predicted_iters_per_s": round(iters / max(walls), 1)}), flush=True)
