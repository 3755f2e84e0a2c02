# kernel trace of a config-4 solve split at iteration 105: which kernels carry the steady phase
cd /tmp && export TMPDIR=/tmp
export SCS_HIP_PSD_COOP=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/c4steady
mkdir -p $O
cat > $O/run.py <<'PY'
import os, sys
sys.path[:0] = [".", "scs-python_amd"]
import scs, problem_gen as pg
from scs import _scs_hip
K, n, k, seed = pg.workload("config4_psd")
d = pg.gen_feasible(K, n, k, seed, lambda z, K: _scs_hip.proj_cone(z, K, dual=True))[0]
s = scs.SCS(d, K, verbose=False, eps_abs=0., eps_rel=0., eps_infeas=0., max_iters=int(sys.argv[1]))
r = s.solve()
print("iters", r["info"]["iter"], "solve ms", r["info"]["solve_time"], "cone ms", r["info"]["cone_time"], "lin ms", r["info"]["lin_sys_time"], "accel", r["info"]["accel_time"])
PY
for it in 105 225; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace$it -o run -- python3 $O/run.py $it > $O/log$it.txt 2>&1
tail -1 $O/log$it.txt
python3 tools/rocpd_summary.py $(find $O/trace$it -name "*.db" | head -1) > $O/summary$it.txt 2>&1
head -14 $O/summary$it.txt | cut -c1-175
done
