"""where a config-5 member's construction spends its host time: the Python front end against the C call (cProfile, one thread)"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import scs, problem_gen as pg
from scs import _scs_hip
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
K, n, k, seed = pg.workload("config5_small")
LS = scs.LinearSolver.HIP_DENSE if (len(sys.argv) > 1 and sys.argv[1] == "dense") else scs.LinearSolver.HIP_INDIRECT
datas = [pg.gen_feasible(K, n, k, seed + i, proj)[0] for i in range(72)]
keep = [scs.SCS(d, K, linear_solver=LS, verbose=False) for d in datas[:8]]
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
keep += [scs.SCS(d, K, linear_solver=LS, verbose=False) for d in datas[8:]]
pr.disable()
print("64 constructions: %.2f ms each" % ((time.perf_counter() - t) * 1e3 / 64))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
