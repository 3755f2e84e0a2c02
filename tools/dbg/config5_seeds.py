"""which seeds of the 512-problem config-5 batch take the fewest / median / most iterations (default settings), per linear solver"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import scs, problem_gen as pg
from scs import _scs_hip
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
Kb, nb_, kb_, seedb = pg.workload("config5_small")
datas = [pg.gen_feasible(Kb, nb_, kb_, seedb + i, proj)[0] for i in range(512)]
for ls in ("hip_indirect", "hip_dense"):
    res = scs.solve_batch([scs.SCS(d, Kb, verbose=False, linear_solver=ls) for d in datas])
    order = sorted(range(512), key=lambda i: res[i]["info"]["iter"])
    pick = [order[0], order[256], order[-1]]
    print(ls, "min / median / max:", [(seedb + i, res[i]["info"]["iter"]) for i in pick], "top5", [(seedb + i, res[i]["info"]["iter"]) for i in order[-5:]])
