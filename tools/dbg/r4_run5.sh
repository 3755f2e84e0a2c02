mkdir -p gpurun_out/r4e
timeout 1200 python -m pytest tests/test_dense_gpu.py -q 2>&1 | tail -5 | tee gpurun_out/r4e/dense_tests.txt
LINSYS=hip_dense PROF_TIMEOUT=400 bash tools/prof_py.sh r4e_batch tools/batch_leg.py 512 16 1
cp gpurun_out/py_r4e_batch/summary.txt gpurun_out/r4e/batch_dense_kernels.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r4e/setup_timing.txt
import os, sys, time
os.environ["SCS_HIP_SETUP_TIMING"] = "1"
sys.path[:0] = [".", "scs-python_amd"]
import scs, problem_gen as pg
from scs import _scs_hip
K, n, k, seed = pg.workload("config5_small")
d = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))[0]
scs.SCS(d, K, verbose=False, linear_solver="hip_dense")
print("---- second init")
t = time.perf_counter(); s = scs.SCS(d, K, verbose=False, linear_solver="hip_dense"); print("init %.2f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); del s; print("finish %.2f ms" % ((time.perf_counter() - t) * 1e3))
PY
