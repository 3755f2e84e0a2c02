"""config-5 batch leg alone (bench.py's `config5_batch`), with the phases timed: python tools/batch_leg.py [N] [threads] [grouped 0/1]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import numpy as np
import torch
import scs
from scs import _scs_hip, batch as scs_batch
import problem_gen as pg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
grouped = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
Kb, nb_, kb_, seedb = pg.workload("config5_small")
t = time.perf_counter()
extra = {"max_iters": int(os.environ["MAXIT"])} if os.environ.get("MAXIT") else {}
if os.environ.get("NORMALIZE"):
    extra["normalize"] = bool(int(os.environ["NORMALIZE"]))   # lab: NORMALIZE=0 shows what the equilibration launches cost scs_init
if os.environ.get("LINSYS"):
    extra["linear_solver"] = os.environ["LINSYS"]   # hip_dense / hip_indirect
problems = [(pg.gen_feasible(Kb, nb_, kb_, seedb + i, proj)[0], Kb, dict(verbose=False, **extra)) for i in range(N)]
tgen = time.perf_counter() - t
scs.SCS(problems[0][0], Kb, verbose=False, max_iters=50, **({"linear_solver": extra["linear_solver"]} if "linear_solver" in extra else {})).solve()
torch.cuda.synchronize()
timing = {}
t = time.perf_counter()
res = scs_batch.solve_sharded(problems, threads=threads, grouped=grouped, timing=timing)
torch.cuda.synchronize()
wall = time.perf_counter() - t
its = sum(r["info"]["iter"] for r in res)
ok = sum(r["info"]["status_val"] == 1 for r in res)
iters = sorted(r["info"]["iter"] for r in res)
print(json.dumps({"problems": N, "grouped": grouped, "threads": threads, "wall_s": round(wall, 3), "iters_per_s": round(its / wall, 1),
                  "solved": ok, "total_iters": its, "gen_s": round(tgen, 2), "timing": {k: round(v, 3) for k, v in timing.items()},
                  "iter_quantiles": [iters[0], iters[len(iters) // 4], iters[len(iters) // 2], iters[3 * len(iters) // 4], iters[-1]],
                  "lin_sys": res[0]["info"].get("lin_sys_solver", "")[:40], "cg_per_iter": round(sum(r["info"]["cg_iters"] for r in res) / its, 2)}))
