"""K9 micro-benchmark: batched PSD projections through the C-ABI (cold eigen-solves: the one-shot entry)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scs-python_amd")); sys.path.insert(0, ROOT)
from scs import _scs_hip as hip
rng = np.random.default_rng(0)
for n, cnt in ((20, 256), (64, 64), (200, 50)):
    K = {"s": [n] * cnt}
    m = cnt * n * (n + 1) // 2
    z = rng.standard_normal(m)
    hip.proj_cone(z, K)
    t = time.time()
    for _ in range(3):
        hip.proj_cone(z, K)
    print("n", n, "count", cnt, "wall per call (incl. H2D/D2H) %.2f ms" % ((time.time() - t) / 3 * 1e3))
