"""scs_init / scs_finish cost of one config-5 problem (debugging aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import numpy as np
import scs
from scs import _scs_hip
import problem_gen as pg
proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
Kb, nb_, kb_, seedb = pg.workload("config5_small")
probs = [pg.gen_feasible(Kb, nb_, kb_, seedb + i, proj)[0] for i in range(64)]
s0 = scs.SCS(probs[0], Kb, verbose=False)
t = time.perf_counter()
sv = [scs.SCS(d, Kb, verbose=False) for d in probs]
t1 = time.perf_counter()
print("init: %.2f ms each (single thread, 64 problems)" % ((t1 - t) * 1e3 / 64))
del sv
t2 = time.perf_counter()
print("finish: %.2f ms each" % ((t2 - t1) * 1e3 / 64))
os.environ["SCS_HIP_SETUP_TIMING"] = "1"
scs.SCS(probs[1], Kb, verbose=False)
