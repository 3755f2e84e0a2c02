"""config 2: CG steps per queued iteration against the chunk that was enqueued for it (SCS_HIP_DEBUG=pipe prints both on stderr)."""
import os, sys, time, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "scs-python_amd")]
    import numpy as np, scs
    from scs import _scs_hip
    import problem_gen as pg
    proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)
    wl = sys.argv[2]
    K, n, k, seed = pg.workload(wl)
    data, _, _ = pg.gen_feasible(K, n, k, seed, proj, pattern=pg.workload_pattern(wl))
    common = dict(linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False, acceleration_lookback=10)
    scs.SCS(data, K, max_iters=10, **common).solve()
    for it in (110, 110, 400):
        s = scs.SCS(data, K, max_iters=it, **common)
        t = time.perf_counter(); s.solve(); el = time.perf_counter() - t
        print("RATE %d iters: %.1f iters/s" % (it, it / el), flush=True)
    sys.exit(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "config2_lp_soc"
env = dict(os.environ, SCS_HIP_DEBUG="pipe")
p = subprocess.run([sys.executable, __file__, "child", wl], env=env, capture_output=True, text=True)
steps = [int(m.group(2)) for m in re.finditer(r"iter (\d+): (\d+) CG steps", p.stderr)]
stalls = len(re.findall(r"STALL", p.stderr))
print(p.stdout)
print("stalls", stalls, "n", len(steps))
print("last 400-iteration solve:", steps[-400:])
p = subprocess.run([sys.executable, __file__, "child", wl], env=os.environ, capture_output=True, text=True)
print("without the debug switch:", p.stdout)
