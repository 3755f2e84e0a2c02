mkdir -p gpurun_out/r4f
timeout 1200 python -m pytest tests/test_dense_gpu.py -q 2>&1 | tail -4 | tee gpurun_out/r4f/dense_tests.txt
LINSYS=hip_dense timeout 600 python tools/batch_leg.py 512 16 1 2>&1 | tail -1 | tee gpurun_out/r4f/batch_dense.txt
timeout 600 python tools/batch_leg.py 512 16 1 2>&1 | tail -1 | tee gpurun_out/r4f/batch_indirect.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r4f/batch_twice.txt
import os, sys, time, json
sys.path[:0] = [".", "scs-python_amd"]
import numpy as np, torch, scs, problem_gen as pg
from scs import _scs_hip, batch as scs_batch
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
Kb, nb_, kb_, seedb = pg.workload("config5_small")
for ls in ("hip_dense", "hip_indirect"):
    problems = [(pg.gen_feasible(Kb, nb_, kb_, seedb + i, proj)[0], Kb, dict(verbose=False, linear_solver=ls)) for i in range(512)]
    for rep in range(2):
        timing = {}
        torch.cuda.synchronize(); t = time.perf_counter()
        res = scs_batch.solve_sharded(problems, threads=16, grouped=True, timing=timing)
        torch.cuda.synchronize(); wall = time.perf_counter() - t
        its = sum(r["info"]["iter"] for r in res)
        print(ls, "rep", rep, "wall %.3f s  %.0f iters/s  init %.3f solve %.3f  solved %d" % (wall, its / wall, timing["init_s"], timing["solve_s"], sum(r["info"]["status_val"] == 1 for r in res)))
PY
