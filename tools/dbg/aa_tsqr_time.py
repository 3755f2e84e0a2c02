"""level-1 TSQR launch at the metric size (l = 3e6, lookback 10): run under tools/prof_py.sh for the kernel durations"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
import numpy as np
from scs import _scs_hip as hip
dim = int(os.environ.get("DIM", "3000001"))
type1 = os.environ.get("TYPE1", "1") == "1"
rng = np.random.RandomState(0)
d = rng.uniform(0.0, 0.9, dim); b = rng.randn(dim)
F = lambda x: b + d * x + 0.05 * np.roll(x, 1)
h = hip.AndersonAccelerator(dim, 10, type1=type1)
x = rng.randn(dim)
for k in range(16):
    f = F(x)
    nrm, fh = h.apply(f, x)
    x = fh
print("ok", h.stats())
