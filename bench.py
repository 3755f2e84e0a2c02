#!/usr/bin/env python3
"""bench.py — ADMM iterations/sec of the MI355X-native SCS hot path + SpMV roofline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  * N > 1 without a launcher (WORLD_SIZE unset): this process starts N ranks itself
    (`python -m torch.distributed.run --nproc-per-node N ... bench.py`, rendezvous on 127.0.0.1) BEFORE it
    touches the GPU, forwards rank 0's JSON line and exits with the children's code.  Under torchrun
    (WORLD_SIZE set) it is one of the ranks; WORLD_SIZE != --gpus is an error.
  * workload (per GPU): the configuration BASELINE.json's metric is quoted on — random LP+SOC cone program
    m=2e6, n=1e6, nnz~2e7 (problem_gen.workload("target_lp_soc")), synthetic, seeded (seed + rank).
  * a "step" = one ADMM iteration of the reference's hot path (KKT solve by PCG over A/A' on the device, cone
    projection, dual update, AA every 10th, convergence check every 25th).  The timed region is ONE
    scs.SCS(...).solve() that executes EXACTLY K iterations from a cold start (max_iters=K, eps=0 so the
    termination test can never fire); inputs are resident in HBM (scs_init uploaded them) when it starts.
    Warm-up = a separate solver instance running W iterations on the same data.  Barrier + device sync on both
    sides, MAX over ranks, value = sum of iterations over ranks / that time ("weak" scaling: the path shards across
    independent problems, no data-path collective — SURVEY §8e); the solutions are then collected with one RCCL
    gather outside the timed region (payload filled device-to-device from the solver's HBM buffers).
  * The iteration rate depends on how many CG steps an iteration needs (12 in the first 20 iterations of a cold
    start, 8 later), so the line also carries the step-count-independent figures `cg_steps_per_s` / `ms_per_cg_step`
    and a second window, `steady_window`: iterations [105, 225) of one longer solve (timestamp taken inside the
    solve), in which the Anderson extrapolations and their safeguards run (the first one fires at iteration 110).
  * roofline: dominant kernel = the CG-step SpMV pair, average launch duration measured live with HIP events on
    the solver's own stream (scs_hip_kernel_times / scs_hip_time_matvec).  `traffic` comes from the committed
    rocprofv3 --pmc passes on the same matrix shape (profiles/), never from this process: `traffic_source` says so.
  * cpu_baseline (rank 0, N=1), all observed in THIS run on this box's host cores: the oracle's CPU-CG variant ("port")
    on the same instance for the first few iterations with 1 thread (in-process) and with all cores (OpenMP timing
    build, child process), and the oracle's sparse-LDL' direct variant ("the QDLDL path") on a ladder of LP sizes,
    every rung a child process with a wall-clock cap — the largest rung that finished and the first that did not.
  * config5_batch: BASELINE.json configs[4] — 512 independent small cone programs sharded round-robin over the
    ranks, each rank's shard ONE grouped solve (scs.solve_batch: the problems share every kernel launch), one
    gather of the solutions; aggregate ADMM iters/s.
  * other_configs (N=1): one bench line each for BASELINE.json configs[1..3] (config 3 with its box cone, config 4
    with the MFMA roofline of the batched PSD projection), plus two lines at the metric workload's size with other sparsity
    patterns: power-law row lengths (layout robustness) and banded (the locality ceiling of K1 / K2, always the LAST line).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "scs-python_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK = 8000.0      # GB/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F64_PEAK = 78.6   # TFLOP/s, MI355X fp64 matrix rate (spec, dense)
STEADY_MARK, STEADY_SPAN = 105, 120


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # the driver's values
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="target_lp_soc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=20)
    ap.add_argument("--cpu-cap-s", type=float, default=30.0, help="wall-clock cap of every CPU child (LDL' rungs, all-core CG)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="OpenMP threads of the multi-core CPU-CG leg (0: min(32, cores) — the best of 8..256 on the 2 x 64-core "
                         "bench host, profiles/r03_cpu_threads.txt)")
    ap.add_argument("--no-steady", action="store_true",
                    help="skip the AA-inclusive window (iterations [%d, %d))" % (STEADY_MARK, STEADY_MARK + STEADY_SPAN))
    ap.add_argument("--no-batch", action="store_true", help="skip the config-5 batch leg")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the lines of BASELINE configs 2-4")
    ap.add_argument("--batch-problems", type=int, default=512)
    ap.add_argument("--batch-threads", type=int, default=0,
                    help="workspaces set up concurrently (scs_init) per rank; 0 = max(1, CPU quota of the box // ranks)")
    ap.add_argument("--batch-one-linsys", action="store_true", help="config-5 leg: only the --batch-linsys solver, not both")
    ap.add_argument("--batch-linsys", default="hip_dense", choices=["hip_dense", "hip_indirect"],
                    help="linear solver of the config-5 members: dense direct (explicit inverse of the reduced KKT matrix, n = 1350) or the indirect PCG path")
    ap.add_argument("--batch-ungrouped", action="store_true", help="config-5 leg as one problem per stream (round 1-2 mode)")
    # testing aids (the driver never passes these): run the N>1 flow on a 1-GPU box
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--force-device", type=int, default=None)
    return ap.parse_args()


def spmv_bytes(nnz, rows, cols):
    """algorithmic bytes of one y = M v: fp64 values, int32 indices"""
    return 12 * nnz + 4 * (rows + 1) + 8 * cols + 8 * rows


def launch_ranks(args):
    """--gpus N without a launcher: become the launcher.  Nothing here has touched the GPU (no torch import, no HIP
    call), the ranks are fresh child processes, and this process only forwards their output and exit code.  The
    rendezvous port is found by bind-and-close; if another process takes it in between the ranks fail at once and
    the launch is repeated on a new port."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = 1
    for _ in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        t0 = time.time()
        rc = subprocess.call(cmd, env=env)
        if rc == 0 or time.time() - t0 > 30:
            break
    return rc


def cpu_quota():
    """CPUs this process may actually use: the cgroup CFS quota (cpu.max) and the affinity mask, whichever is smaller.  Round 4: the
    bench boxes report 256 logical CPUs but run under cpu.max = 1600000 100000 = 16 CPUs — every thread count above that is throttled,
    which is what rounds 2-3 read as 'more threads are slower' (and blamed on first-touch placement)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(round(float(q) / float(per)))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, int(round(q / per))))
    except (OSError, ValueError):
        pass
    return n


def run_child(cmd, cap_s, env=None, partial_ok=False):
    """(json dict | None, seconds, timed_out) of a CPU child process printing JSON lines (the last one counts).  partial_ok: a child
    that runs into the cap is killed and the best `iters_per_s` line it had printed so far is returned (thread sweeps)"""
    t0 = time.perf_counter()
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
    timed_out = False
    try:
        so, _ = pr.communicate(timeout=cap_s)
    except subprocess.TimeoutExpired:
        pr.kill()
        so, _ = pr.communicate()
        timed_out = True
        if not partial_ok:
            return None, time.perf_counter() - t0, True
    dt = time.perf_counter() - t0
    lines = [json.loads(ln) for ln in (so or "").splitlines() if ln.startswith("{")]
    if not lines:
        return None, dt, timed_out
    if timed_out:  # no "best" line was printed: pick it here
        best = max(lines, key=lambda d: d.get("iters_per_s", 0.0))
        return dict(best, sweep=[[d["threads"], round(d["iters_per_s"], 3)] for d in lines], sweep_cut_short=True), dt, True
    return lines[-1], dt, False


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without a launcher)"
                         % (args.gpus, world, args.gpus))

    # before ANY HIP runtime is loaded (torch's comes first): the same default scs._scs_hip sets — see its _runtime_env
    if os.environ.get("SCS_HIP_RUNTIME_ENV", "1") != "0":
        os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1000000")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (exported on the GPU boxes already: RCCL between ranks needs dmabuf IPC there)
    # the library's device block pool (csrc/common.hpp DevPool) keeps 1 GiB by default; this process owns its GPU and tears down 512
    # workspaces at a time in the batch leg: the cap of round 4 (an existing value wins)
    os.environ.setdefault("SCS_HIP_POOL_MB", "16384")
    import torch  # first: its bundled HIP runtime must be the one in the process
    import torch.distributed as dist
    import numpy as np
    import scs
    from scs import _scs_hip
    from scs import batch as scs_batch
    import problem_gen as pg

    if _scs_hip.device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device; the product has no CPU fallback")
    dev = local_rank if args.force_device is None else args.force_device
    torch.cuda.set_device(dev)
    _scs_hip.set_device(dev)
    coll_dev = torch.device("cuda", dev) if args.dist_backend == "nccl" else torch.device("cpu")
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max_sum(t_local, n_local):
        tm = torch.tensor([t_local], dtype=torch.float64, device=coll_dev)
        ns = torch.tensor([float(n_local)], dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dist.all_reduce(ns, op=dist.ReduceOp.SUM)
        return float(tm.item()), float(ns.item())

    proj = lambda z, K: _scs_hip.proj_cone(z, K, dual=True)  # noqa: E731
    common = dict(eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False,
                  acceleration_lookback=10, linear_solver=scs.LinearSolver.HIP_INDIRECT)

    def pmc_traffic(workload, which):
        """HBM bytes of K1 / K2 per launch from the committed rocprofv3 --pmc passes on this workload's matrix shape"""
        for fname in ("r06_pmc_traffic.json", "r06_pmc_traffic_qp.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    pmc = json.load(f)
                if pmc.get("workload") == workload:
                    kk = pmc[which]
                    return int((2 * kk["FETCH_SIZE_KiB"] + kk["WRITE_SIZE_KiB"]) * 1024), "profiles/" + fname
            except (OSError, KeyError, ValueError):
                pass
        return None, None

    def cone_summary(K):
        out = {}
        for kk, vv in K.items():
            if isinstance(vv, list):
                out[kk] = "%dx%s" % (len(vv), vv[0]) if kk in ("q", "s", "cs") else "%d values" % len(vv)
            else:
                out[kk] = vv
        return out

    def measure(workload, steps, warmup, steady, gather):
        """one bench line: exactly `steps` ADMM iterations of `workload` from a cold start, timed as the contract says"""
        K, n, k, seed = pg.workload(workload)
        t0 = time.perf_counter()
        if pg.workload_qp(workload):  # K3: a random PSD-by-construction P rides along (R:scs/py/__init__.py:163-166 hands over triu(P))
            data, p_star, _ = pg.gen_feasible_qp(K, n, k, seed + rank, proj, b_per_col=pg.workload_qp(workload))
        else:
            data, p_star, _ = pg.gen_feasible(K, n, k, seed + rank, proj, pattern=pg.workload_pattern(workload))
        m = data["A"].shape[0]
        nnz = int(data["A"].nnz)
        t_gen = time.perf_counter() - t0
        # (the warm-up workspace is released only AFTER the timed region: freeing 1.5 GB of HBM leaves the runtime with
        # deferred work that the first device-wide synchronize afterwards pays for — 25 ms measured — and that
        # synchronize is the one closing the timed region)
        wsolver = None
        if warmup > 0:
            wsolver = scs.SCS(data, K, max_iters=warmup, **common)
            wsolver.solve()
        solver = scs.SCS(data, K, max_iters=steps, **common)
        if not os.environ.get("BENCH_NO_INSITU"):
            solver._solver._set_profiling(True)
        # Quiet start: some 0.1-0.3 s after host pages that the runtime had pinned for a copy are released (the previous
        # workload's matrix, the generator's temporaries) the kernel driver evicts this process's queues for 30-80 ms — one
        # hole in the middle of whatever runs then; a 40 ms config-2 solve took 115 ms in ~40 % of the default runs
        # (~10 % with GPU_PINNED_MIN_XFER_SIZE raised as above; profiles/r03_queue_eviction.txt).  Everything of this
        # workload is allocated and warmed up by now: let that pass before the clock starts.
        torch.cuda.synchronize()
        time.sleep(0.4)
        # ... and the 0.4 s of idle time let the GPU drop its clocks: ~30 ms of plain device work (no host memory involved, nothing of the
        # solver touched) right in front of the timed region bring them back (a 20-iteration window is 63 ms: one box of the round's last
        # four measured 69 ms = 282 iters/s with normal K1 / K2 rates in the same run, the other three 314-318)
        if warmup > 0:
            spin = torch.ones(1 << 24, dtype=torch.float64, device=torch.device("cuda", dev))
            for _ in range(400):
                spin.mul_(1.0000001)
            torch.cuda.synchronize()
            del spin
        # ---------------- timed region: exactly K ADMM iterations ----------------
        barrier()
        t0 = time.perf_counter()
        sol = solver.solve(warm_start=False)
        t_a = time.perf_counter()
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        barrier()
        elapsed = time.perf_counter() - t0
        if os.environ.get("BENCH_DEBUG_TIMING"):
            print("timed region: solve() %.1f ms, synchronize %.1f ms, barrier %.1f ms" % ((t_a - t0) * 1e3, (t_b - t_a) * 1e3,
                  (time.perf_counter() - t_b) * 1e3), file=sys.stderr)
        info = sol["info"]
        assert info["iter"] == steps, (info["iter"], steps, info["status"])
        del wsolver
        psd_t = solver._solver._time_psd(reps=20)    # None unless the workload has PSD cones
        kt = solver._solver._kernel_times()          # in-situ samples (one CG step per host sync)
        kb = solver._solver._time_matvec(reps=30)    # back-to-back batch, event overhead amortised
        elapsed_max, total_iters = reduce_max_sum(elapsed, info["iter"])
        _, total_cg = reduce_max_sum(elapsed, info["cg_iters"])

        # ---------------- single RCCL gather of the solutions (outside the timed region) ----------------
        gather_ms = None
        if gather and world > 1:
            if args.dist_backend == "nccl":  # the payload never visits the host: x | y | s straight from the solver's HBM buffers
                payload = torch.empty(n + 2 * m, dtype=torch.float64, device=coll_dev)
                base = payload.data_ptr()
                solver._solver.solution_to_device(base, base + 8 * n, base + 8 * (n + m))
            else:
                payload = torch.from_numpy(np.concatenate([sol["x"], sol["y"], sol["s"]])).to(coll_dev)
            bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
            torch.cuda.synchronize()
            tg = time.perf_counter()
            dist.gather(payload, bufs, dst=0)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - tg) * 1e3
            del bufs, payload
        del solver

        # ---------------- steady window: Anderson steps inside ----------------
        steady_out = None
        if steady:
            mark, span = STEADY_MARK, STEADY_SPAN  # the Anderson solves of iterations 110 and 220 (and their safeguards) are inside
            ssolver = scs.SCS(data, K, max_iters=mark + span, **common)
            ssolver._solver._set_mark(mark)
            # (quiet start as above: the workspaces of the timed region were just released — round 4: config 4's window read 223 iters/s
            #  in a default run and 434-451 in five runs of tools/dbg/c4_steady_time.py: one queue-eviction hole of ~0.25 s inside a 0.28 s window)
            torch.cuda.synchronize()
            time.sleep(0.4)
            barrier()
            ssol = ssolver.solve(warm_start=False)
            sinfo, mk = ssol["info"], ssolver._solver._get_mark()
            win_ms = sinfo["solve_time"] - mk["ms"]
            win_cg = sinfo["cg_iters"] - mk["cg_iters"]
            w_max, w_iters = reduce_max_sum(win_ms * 1e-3, span)
            _, w_cg = reduce_max_sum(0.0, win_cg)
            steady_out = {
                "window": "ADMM iterations [%d, %d) of one cold-started solve (timestamp inside scs_solve, stream drained)" % (mark, mark + span),
                "value": round(w_iters / w_max, 3), "unit": "ADMM iters/s", "ms_per_step": round(w_max * 1e3 / span, 4),
                "cg_steps_per_admm_iter": round(w_cg / w_iters, 2), "cg_steps_per_s": round(w_cg / w_max, 1),
                "aa_calls_in_window": sinfo["aa_stats"]["iter"] - mk["aa_calls"],
                "aa_accepted_in_window": sinfo["aa_stats"]["n_accept"] - mk["aa_accept"],
                "aa_safeguard_rejects_total": sinfo["aa_stats"]["n_safeguard_reject"],
                "accel_ms_total": round(sinfo["accel_time"], 2),
            }
            del ssolver

        # ---------------- roofline of the dominant kernel ----------------
        # Two live HIP-event measurements on the solver's stream.  (i) in-situ: single launches inside
        # the timed solve, each bracketed by its own event pair — carries ~15-25 us of event/dispatch
        # overhead per sample; (ii) batch: 30 back-to-back launches per event pair on the same resident
        # data right after the solve.  (ii) is the kernel's launch duration (it is what rocprofv3
        # --kernel-trace reports, profiles/); (i) is kept as a cross-check.
        k1_situ = kt["k1_ms"] / max(kt["k1_n"], 1)
        k2_situ = kt["k2_ms"] / max(kt["k2_n"], 1)
        k1_avg, k2_avg = kb["k1_ms"], kb["k2_ms"]
        b1 = spmv_bytes(nnz, m, n)              # SURVEY 8(d) formula (R_y comes as two scalars: nothing else is read)
        b2 = spmv_bytes(nnz, n, m) + 8 * n      # + p read by the fused epilogue (R_x is a scalar)
        gb1 = b1 / (k1_avg * 1e-3) / 1e9 if k1_avg > 0 else 0.0
        gb2 = b2 / (k2_avg * 1e-3) / 1e9 if k2_avg > 0 else 0.0
        lss = info.get("lin_sys_solver", "")
        kname = ("k_spmv_cs_il" if os.environ.get("SCS_HIP_CS_SCHED", "3") in ("2", "3") else "k_spmv_cs_ga") if "column-sorted" in lss \
            else "k_spmv_slab" if "slab" in lss else "k_spmv_stream"
        k1_dom = k1_avg >= k2_avg
        dom = ("K1 %s<EpiDivR> (z = R_y^-1 A p)" % kname, b1, k1_avg, gb1) if k1_dom else \
              ("K2 %s<EpiGp> (Gp = A'z + R_x p)" % kname, b2, k2_avg, gb2)
        traffic, traffic_src = pmc_traffic(workload, "K1" if k1_dom else "K2")
        roofline = {
            "bound": "hbm", "achieved": round(dom[3], 1), "peak": HBM_PEAK, "unit": "GB/s",
            "frac": round(dom[3] / HBM_PEAK, 4), "traffic": traffic,
            "traffic_source": ("%s (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of the same kernel on this workload's "
                               "matrices, collected in a separate committed run — NOT measured by this process)" % traffic_src)
            if traffic is not None else None,
            "kernel": dom[0], "algorithmic_bytes_per_launch": int(dom[1]), "avg_launch_ms": round(dom[2], 5),
            "samples": kt["k1_n"],
            "k1": {"bytes": int(b1), "avg_ms": round(k1_avg, 5), "GBps": round(gb1, 1), "frac": round(gb1 / HBM_PEAK, 4),
                   "in_situ_event_ms": round(k1_situ, 5)},
            "k2": {"bytes": int(b2), "avg_ms": round(k2_avg, 5), "GBps": round(gb2, 1), "frac": round(gb2 / HBM_PEAK, 4),
                   "in_situ_event_ms": round(k2_situ, 5)},
        }
        if "P" in data:
            # K3 (QP path): Gp = P p.  Algorithmic bytes per SURVEY §2.1 K3 — the caller hands over triu(P) and "each stored entry
            # contributes twice": 12 B per STORED entry + row pointers + p in + Gp out.  The kernel streams the full symmetric CSR
            # (csrc/scs_hip.hip Pf: 2 nnz - n entries) — an atomics-free, order-fixed product needs every value in row order of both
            # triangles (profiles/r06_k3.txt), so `frac` against the algorithmic bytes is about half of `frac_streamed`.
            nnz_pu = int(data["P"].nnz)
            nnz_pf = 2 * nnz_pu - n
            b3 = 12 * nnz_pu + 4 * (n + 1) + 8 * n + 8 * n
            b3s = 12 * nnz_pf + 4 * (n + 1) + 8 * n + 8 * n
            # three live measurements on the solver's stream: (i) in situ — the sampled CG steps of the timed solve carry a fourth event behind
            # K3 (what rocprofv3 sees: profiles/r06_k3.txt; ~3 us of event overhead per sample); (ii) the marginal cost of K3 in a CG step,
            # (K1, K3, K2) x 30 minus (K1, K2) x 30 — it also pays for the lines K3 pushes out of the caches in front of K2; (iii) back to back:
            # the 224 MB of Pf then stay in the 256 MB Infinity Cache between launches (flattering)
            k3_avg = kt["k3_ms"] / max(kt["k3_n"], 1) if kt.get("k3_n", 0) > 0 else kb["k3_ms"]
            gb3 = b3 / (k3_avg * 1e-3) / 1e9 if k3_avg > 0 else 0.0
            roofline["k3"] = {"kernel": "K3 %s<EpiStore> (Gp = P p) on the full symmetric CSR of P" % kname, "nnz_triu_P": nnz_pu,
                              "bytes": int(b3), "avg_ms": round(k3_avg, 5), "GBps": round(gb3, 1), "frac": round(gb3 / HBM_PEAK, 4),
                              "streamed_bytes": int(b3s), "frac_streamed": round(b3s / (k3_avg * 1e-3) / 1e9 / HBM_PEAK, 4) if k3_avg > 0 else 0.0,
                              "samples": int(kt.get("k3_n", 0)), "how": "in situ: HIP events around K3 inside sampled CG steps of the timed solve",
                              "traffic": pmc_traffic(workload, "K3")[0], "traffic_source": pmc_traffic(workload, "K3")[1],
                              "marginal_ms_in_a_cg_step": round(kb["k3_ms"], 5), "avg_ms_back_to_back": round(kb["k3_back_to_back_ms"], 5),
                              "launches_per_cg_step": "K1 + K3 + K2 (K2's epilogue adds P p)"}
        if psd_t is not None:
            # PSD-heavy workloads: the batched eigen-solve (K9) is the dominant kernel and the matrix cores bound it.
            # achieved = reference flop count of a LAPACK-style symmetric eigensolve + reconstruction of the same matrices
            # ((16/3 + 2) n^3 each) / measured time; the Jacobi method spends more (the MFMA instruction count is
            # in profiles/).  duration: in situ — the cone kernels of the timed solve's queued iterations between two HIP
            # events on the solver's stream (K9 is all of it but the one-launch `l` / `q` kernels); the stand-alone
            # re-projection of one iterate (psd_t["ms"]) is the fully warm lower bound, quoted beside it
            cone_ms = kt["cone_ms"] / max(kt["cone_n"], 1)
            if cone_ms > 0:
                psd_t = dict(psd_t, ms_same_vector=round(psd_t["ms"], 4), ms=cone_ms)
            tf = psd_t["ref_flops"] / (psd_t["ms"] * 1e-3) / 1e12
            spmv_roofline = roofline
            roofline = {"bound": "mfma", "achieved": round(tf, 4), "peak": MFMA_F64_PEAK, "unit": "TFLOP/s", "frac": round(tf / MFMA_F64_PEAK, 5),
                        "traffic": None, "traffic_source": None,
                        "kernel": "K9 batched PSD projection (k_psd_sweep_mc + k_psd_gemm + k_psd_apply_v / _q), %d matrices of order <= %d, "
                        "warm-started" % (psd_t["matrices"], psd_t["max_order"]),
                        "algorithmic_flops_per_launch": psd_t["ref_flops"], "avg_launch_ms": round(psd_t["ms"], 4),
                        "samples": kt["cone_n"], "ms_reprojecting_the_same_vector": psd_t.get("ms_same_vector"),
                        "spmv": {k_: spmv_roofline[k_] for k_ in ("k1", "k2")}}
        cg_per_s = total_cg / elapsed_max
        line = {
            "metric": "ADMM iters/sec (random %s cone program, indirect CG linsys, AA lookback 10)" % (
                "LP+SOC" if "lp_soc" in workload else workload),
            "value": round(total_iters / elapsed_max, 4),
            "unit": "ADMM iters/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed_max * 1e3 / steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "%s: m=%d n=%d nnz=%d cone=%s seed=%d(+rank); one independent instance per GPU" % (
                    workload, m, n, nnz, cone_summary(K), seed),
                "world_size_seen": world, "backend": args.dist_backend if world > 1 else None,
                "cg_steps_per_admm_iter": round(info["cg_iters"] / max(info["iter"], 1), 2),
                "cg_steps_per_s": round(cg_per_s, 1), "ms_per_cg_step": round(1e3 / cg_per_s * world, 4),
                "admm_iters_timed": int(info["iter"]), "solve_ms_inside_scs_solve": round(info["solve_time"], 2),
                "aa_extrapolations_in_timed_region": int(info["aa_stats"]["n_accept"]),
                "lin_sys_ms": round(info["lin_sys_time"], 1), "cone_ms": round(info["cone_time"], 1),
                "accel_ms": round(info["accel_time"], 1), "setup_ms": round(info["setup_time"], 1),
                "gen_s": round(t_gen, 1), "gather_ms": gather_ms,
            },
            "steady_window": steady_out,
            "roofline": roofline,
        }
        return line, data, K

    out, data, K = measure(args.workload, args.steps, args.warmup, steady=not args.no_steady, gather=True)

    # ---------------- config 5: batch of independent small problems, sharded over the ranks ----------------
    batch_leg = None
    if not args.no_batch:
        Kb, nb_, kb_, seedb = pg.workload("config5_small")
        NB = args.batch_problems
        mine = set(scs_batch.shard_indices(NB, rank, world))
        mb = pg.cone_dims(Kb)
        tgen = time.perf_counter()
        datas, pstars = {}, {}
        for i in range(NB):  # a rank only generates (and touches) its own shard
            if i in mine:
                datas[i], pstars[i], _ = pg.gen_feasible(Kb, nb_, kb_, seedb + i, proj)
        tgen = time.perf_counter() - tgen
        dims = [(nb_, mb)] * NB
        # setup threads of this rank: the ranks of a node share the box's CPU quota (bench boxes: 16 CPUs under a cgroup, bench.py
        # cpu_quota) — 8 ranks x 16 setup threads + 8 host loops would fight over them
        bthreads = args.batch_threads if args.batch_threads > 0 else max(1, cpu_quota() // max(world, 1))

        def run_batch(ls_name):
            """one sharded batch solve with the members' linear solver `ls_name`; returns the leg's record on rank 0"""
            batch_ls = scs.LinearSolver(ls_name)
            problems = [(datas[i], Kb, dict(verbose=False, linear_solver=batch_ls)) if i in mine else None for i in range(NB)]
            # warm the kernels of this shape once (code objects, allocator)
            if mine:
                scs.SCS(problems[min(mine)][0], Kb, verbose=False, max_iters=50, linear_solver=batch_ls).solve()
            timing = {}
            barrier()
            tb = time.perf_counter()
            res = scs_batch.solve_sharded(problems, dims=dims, threads=bthreads, device=coll_dev,
                                          grouped=not args.batch_ungrouped, timing=timing)
            torch.cuda.synchronize()
            barrier()
            tb = time.perf_counter() - tb
            tb_max, _ = reduce_max_sum(tb, 0)
            if rank != 0:
                return None
            its = sum(r["info"]["iter"] for r in res)
            ok = sum(r["info"]["status_val"] == 1 for r in res)
            iters_sorted = sorted(r["info"]["iter"] for r in res)
            # every problem rank 0 generated (all of them at N = 1) against the optimum its generator constructed: |pobj - p*| / max(1, |p*|)
            # (default settings stop at 1e-4 on the residuals; the objective is then ~1e-3 off — tests/test_group_gpu.py pins x, y, s at 1e-10)
            obj_err = max([abs(res[i]["info"]["pobj"] - pstars[i]) / max(1.0, abs(pstars[i])) for i in mine] or [0.0])
            return {"value": round(its / tb_max, 1), "objective_max_rel_err_vs_constructed_optimum": float("%.2e" % obj_err),
                    "objective_checked": len(mine), "problems_per_s": round(NB / tb_max, 2), "solved": int(ok), "total_iters": int(its),
                    "wall_s": round(tb_max, 3), "iterations_min_median_max": [iters_sorted[0], iters_sorted[len(iters_sorted) // 2], iters_sorted[-1]],
                    "rank0_phases_s": {k_: round(v_, 3) for k_, v_ in timing.items()}, "linear_solver": batch_ls.value}

        # the line's `value` is the leg with --batch-linsys (default: the dense direct solver, csrc/dense.hpp, n = 1350 <= 8192); the OTHER
        # linear solver — the indirect PCG path this repository is about — runs right behind it and rides along
        first = run_batch(args.batch_linsys)
        other_ls = "hip_indirect" if args.batch_linsys == "hip_dense" else "hip_dense"
        second = None if args.batch_one_linsys else run_batch(other_ls)
        if rank == 0:
            mode = ("one problem per stream, %d in flight per GPU" % bthreads) if args.batch_ungrouped else \
                "each rank's shard ONE grouped solve (scs.solve_batch: lock step, every kernel launch shared), %d workspaces set up at a time" % bthreads
            batch_leg = {
                "workload": "config5: %d independent problems, each cone={'l': 2000, 'q': '20x50', 's': '5x20'} m=%d n=%d, seeds %d..%d, "
                            "default settings (eps 1e-4), problem i -> rank i %% %d, %s, one gather of [x|y|s]"
                            % (NB, mb, nb_, seedb, seedb + NB - 1, world, mode),
                "unit": "ADMM iters/s (aggregate, wall time incl. scs_init and the gather)",
                "problems": NB, "gen_s_rank0": round(tgen, 2), "n_gpus": world, "setup_threads_per_rank": bthreads,
            }
            batch_leg.update(first)
            batch_leg["value_is"] = ("members solved with LinearSolver.%s" % args.batch_linsys.upper() +
                                     (" — the dense DIRECT solver (SURVEY §8 f4), not north_star's indirect path: that figure is "
                                      "`value_north_star_path`" if args.batch_linsys == "hip_dense" else " — north_star's indirect PCG path"))
            if args.batch_linsys == "hip_indirect":
                batch_leg["value_north_star_path"] = first["value"]
            if second is not None:
                batch_leg["value_" + other_ls] = second["value"]
                batch_leg["other_linear_solver"] = second
                if other_ls == "hip_indirect":
                    batch_leg["value_north_star_path"] = second["value"]   # the indirect PCG path (csrc/batch.hpp grouped PCG)

    # ---------------- other BASELINE configs: one line each (N = 1) ----------------
    other = None
    if world == 1 and not args.no_other_configs:
        other = []
        for wl, st, wu in (("config2_lp_soc", 100, 10), ("config3_mixed", 20, 3), ("config4_psd", 100, 5), ("target_qp", 20, 2),
                           ("powerlaw_lp", 20, 2), ("banded_lp", 20, 2)):
            if wl == args.workload or (os.environ.get("BENCH_OTHER") and wl not in os.environ["BENCH_OTHER"].split(",")):
                continue
            # config 4: the first 100 iterations run with the residual-tied PSD sweep level, later ones do not — the
            # line carries the window past iteration 100 and a whole solve next to the cold-start window
            line, d4, K4 = measure(wl, st, wu, steady=(wl == "config4_psd"), gather=False)
            if wl == "config4_psd":
                ws4 = scs.SCS(d4, K4, verbose=False, acceleration_lookback=10, linear_solver=scs.LinearSolver.HIP_INDIRECT)
                ws4._solver._set_profiling(True)  # in-situ events around the cone kernels of the queued iterations (as in the timed window)
                torch.cuda.synchronize()
                time.sleep(0.4)
                tw = time.perf_counter()
                sol4 = ws4.solve(warm_start=False)
                torch.cuda.synchronize()
                tw = time.perf_counter() - tw
                line["whole_solve"] = {"settings": "defaults (eps_abs = eps_rel = 1e-4)", "status": sol4["info"]["status"],
                                       "iterations": sol4["info"]["iter"], "wall_s": round(tw, 3),
                                       "value": round(sol4["info"]["iter"] / tw, 1), "unit": "ADMM iters/s over the whole solve",
                                       "cg_steps_per_admm_iter": round(sol4["info"]["cg_iters"] / max(sol4["info"]["iter"], 1), 2)}
                # K9 over the WHOLE solve (the window above is the cold start, where the sweeps dominate): average of the in-situ samples
                kt4 = ws4._solver._kernel_times()
                if kt4["cone_n"] > 0:
                    k9_ms = kt4["cone_ms"] / kt4["cone_n"]
                    ref_fl = sum((16. / 3. + 2.) * float(k_) ** 3 for k_ in K4.get("s", []))
                    line["whole_solve"]["k9"] = {"ms_per_projection": round(k9_ms, 4), "samples": kt4["cone_n"],
                                                 "achieved_tflops": round(ref_fl / (k9_ms * 1e-3) / 1e12, 3),
                                                 "frac_of_fp64_mfma_peak": round(ref_fl / (k9_ms * 1e-3) / 1e12 / MFMA_F64_PEAK, 4),
                                                 "what": "cone kernels of the queued iterations between two HIP events (K9 is all of it but the one-launch l kernel); "
                                                         "flops = the reference count of the roofline entry above"}
                # K9's refinement stage (round 5, csrc/psd.hpp psd_stop_test): how many of the solve's projections took it, per matrix
                st4 = ws4._solver._psd_refine_stats()
                line["whole_solve"]["psd_refinement"] = {
                    "projections_refined_per_matrix_mean_min_max": [round(float(st4[:, 0].mean()), 1), int(st4[:, 0].min()), int(st4[:, 0].max())],
                    "refinements_sent_back_to_the_sweeps_per_matrix_mean": round(float(st4[:, 1].mean()), 2),
                    "what": "GEMM-only removal of the mixed-sign part of V'AV behind the Jacobi sweeps; SCS_HIP_PSD_REFINE=0 restores the strict sweeps"}
                del ws4, sol4
            else:
                line.pop("steady_window", None)
            if wl == "config3_mixed":
                # the 20-step window above is a cold start (333 CG steps per iteration); a WHOLE solve averages half that (round 5)
                ws3 = scs.SCS(d4, K4, verbose=False, acceleration_lookback=10, linear_solver=scs.LinearSolver.HIP_INDIRECT, max_iters=2000)
                torch.cuda.synchronize()
                time.sleep(0.4)
                tw = time.perf_counter()
                sol3 = ws3.solve(warm_start=False)
                torch.cuda.synchronize()
                tw = time.perf_counter() - tw
                line["whole_solve"] = {"settings": "defaults (eps_abs = eps_rel = 1e-4)", "status": sol3["info"]["status"],
                                       "iterations": sol3["info"]["iter"], "wall_s": round(tw, 3),
                                       "value": round(sol3["info"]["iter"] / tw, 1), "unit": "ADMM iters/s over the whole solve",
                                       "cg_steps_per_admm_iter": round(sol3["info"]["cg_iters"] / max(sol3["info"]["iter"], 1), 2),
                                       "krylov": sol3["info"]["lin_sys_solver"],
                                       "minres": "built (csrc/minres.hpp, SCS_HIP_KRYLOV=minres) and measured on this solve: 241 steps per iteration and "
                                                 "825 iterations against PCG's 170 and 700 — 2.1 x slower, not the default (profiles/r05_config3_minres.txt)"}
                del ws3, sol3
                line["config"]["why_so_many_cg_steps"] = (
                    "conditioning, not cone kernels: R_y weighs the 100,000 zero-cone rows 1000 x heavier than the others "
                    "(1/(1000 scale) vs 1/scale), the reduced system's condition number is ~1e3 and Jacobi-preconditioned CG needs "
                    "~330 steps per solve; with the same rows declared `l` it needs 13 (profiles/r03_config3_cg_study.txt)")
            del d4, K4
            if wl == "target_qp":
                line["config"]["why_this_line"] = (
                    "not a BASELINE config: the metric workload's A with a quadratic objective, P = I + B'B (PSD by construction, "
                    "nnz(triu P) ~ 8.5e6) — the QP path: one more product per CG step (K3), roofline.k3 (VERDICT r05 item 4)")
            if wl == "powerlaw_lp":
                line["config"]["why_this_line"] = (
                    "not a BASELINE config: the metric workload's size with Pareto(1.3) row lengths (up to 20 000 nonzeros per row) — "
                    "layout robustness: rows too long for the pass layout's count fields ride in the passes as pieces "
                    "(round 2: summed whole by a side launch, K1 164.6 us, 152.6 iters/s; profiles/r03_patterns.txt)")
            if wl == "banded_lp":
                line["config"]["why_this_line"] = (
                    "not a BASELINE config: the metric workload's size with a banded pattern — the same K1 / K2 kernels when the gathers "
                    "have locality (the ceiling of this decomposition; the uniformly random pattern is bound by L1 line fills, "
                    "profiles/r03_spmv_pmc.txt)")
                if args.workload == "target_lp_soc" and out["roofline"].get("bound") == "hbm":
                    out["roofline"]["locality_ceiling"] = {
                        "frac": line["roofline"]["frac"], "achieved": line["roofline"]["achieved"], "unit": "GB/s",
                        "what": "the same kernel on the banded_lp workload (same m, n, nnz; measured in this run, other_configs[-1])",
                        "bound_of_the_random_pattern": "vector-L1 miss path: TCP_PENDING_STALL 63 % of the launch, 6.4 x algorithmic bytes as "
                                                       "128-byte line fills L2 -> L1, HBM traffic 1.29-1.33 x (profiles/r03_spmv_pmc.txt)"}
            other.append(line)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---------------- CPU baseline (oracle; bounded samples; everything observed in this run) ----------------
    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import scs_oracle  # the checker, timed beside the product; never in the product path
        ncores = os.cpu_count() or 1
        nquota = cpu_quota()
        child = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py")]
        # the "QDLDL path": the oracle's sparse LDL' direct variant on a ladder of LP sizes (the shape of configs[0],
        # m = 2 n, 50 nonzeros per column), every rung a single-threaded child with a wall-clock cap, all started now so
        # that they run beside the in-process leg below (they use 3 of this box's cores)
        ladder = [(4000, 2000), (8000, 4000), (16000, 8000)]
        procs = []
        for (lm, ln) in ladder:
            procs.append((lm, ln, time.perf_counter(),
                          subprocess.Popen(child + ["ldl", str(lm), str(ln), "50", "1", "200"], stdout=subprocess.PIPE,
                                           stderr=subprocess.DEVNULL, text=True)))
        ci = max(1, args.cpu_iters)
        stg = dict(eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0, verbose=False, acceleration_lookback=10, max_iters=ci)
        ref = scs_oracle.solve(data, K, indirect=True, **stg)
        cpu_ms = ref["info"]["solve_time"]
        gsolver = scs.SCS(data, K, max_iters=ci, **common)
        gsol = gsolver.solve(warm_start=False)
        gpu_ms = gsol["info"]["solve_time"]
        del gsolver
        rungs = []
        for lm, ln, t0, pr in procs:
            left = max(0.0, args.cpu_cap_s - (time.perf_counter() - t0))
            try:
                so, _ = pr.communicate(timeout=left)
                js = [json.loads(x) for x in so.splitlines() if x.startswith("{")]
                rungs.append(dict(js[-1], finished=True) if js else {"m": lm, "n": ln, "finished": False, "error": "no output"})
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.communicate()
                rungs.append({"m": lm, "n": ln, "finished": False, "cap_s": args.cpu_cap_s})
        done = [r for r in rungs if r.get("finished")]
        notdone = [r for r in rungs if not r.get("finished")]
        hip_same = None
        if done:  # the HIP path on the largest LP the direct variant finished
            big = done[-1]
            dK = {"l": big["m"]}
            dd, _, _ = pg.gen_feasible(dK, big["n"], 50, 1, proj)
            dgpu = scs.SCS(dd, dK, linear_solver=scs.LinearSolver.HIP_INDIRECT, eps_abs=0.0, eps_rel=0.0, eps_infeas=0.0,
                           verbose=False, acceleration_lookback=10, max_iters=200).solve(warm_start=False)
            hip_same = round(200 / (dgpu["info"]["solve_time"] * 1e-3), 1)
        # all cores: the OpenMP timing build of the same CPU-CG variant on the same workload (child process)
        # thread sweep inside ONE child (the problem is generated once; every count gets a fresh workspace whose pages are first
        # touched by its own threads: oracle/oscs.h): a quarter, half and all of the logical CPUs; the best one is the figure
        if args.cpu_threads > 0:
            sweep = [args.cpu_threads]
        else:
            # around what the cgroup lets this process use (cpu_quota): half of it, all of it, twice it
            sweep = sorted({max(1, nquota // 2), nquota, min(ncores, 2 * nquota)})
        allc, allc_s, allc_to = run_child(child + ["cg", args.workload, str(ci), ",".join(str(t) for t in sweep)], max(args.cpu_cap_s, 60.0) * len(sweep),
                                          partial_ok=True)
        nthr = allc["threads"] if allc else sweep[-1]
        cpu_baseline = {
            "value": round(ci / (cpu_ms * 1e-3), 5), "unit": "ADMM iters/s", "cores": 1, "kind": "port",
            "sample": "first %d ADMM iterations (cold start, %d CG steps) of the same instance on the oracle's "
                      "CPU-CG variant: %.1f s; the HIP path runs the same %d iterations (%d CG steps) in %.3f s; "
                      "host has %d logical CPUs, %d usable by this process" % (ci, ref["info"]["cg_iters"], cpu_ms * 1e-3, ci, gsol["info"]["cg_iters"],
                                             gpu_ms * 1e-3, ncores, nquota),
            "multi_core": ({"value": round(allc["iters_per_s"], 4), "unit": "ADMM iters/s", "cores": nthr, "host_cores": ncores,
                            "cpus_this_process_may_use": nquota, "kind": "port",
                            "thread_sweep_iters_per_s": allc.get("sweep", [[nthr, round(allc["iters_per_s"], 3)]]),
                            "sample": "the same %d iterations (%d CG steps) with the OpenMP build of the oracle (row- / column-parallel "
                                      "mat-vecs, parallel vector loops, parallel l / SOC / exp projections; Anderson steps sequential; vectors and "
                                      "matrix copies first touched by the threads that stream them, OMP_PROC_BIND=spread) on %d threads, the best "
                                      "of the sweep: %.2f s.  The host has %d logical CPUs but the cgroup of this process allows %d (cpu.max): "
                                      "thread counts beyond that are throttled, not faster" % (ci, allc["cg_steps"], nthr, allc["solve_s"], ncores, nquota)}
                           if allc else {"value": None, "cores": nthr, "host_cores": ncores,
                                         "sample": "did not finish within %.0f s" % max(args.cpu_cap_s, 60.0) if allc_to else "child failed"}),
            "direct_ldl": {
                "what": "oracle's sparse LDL' direct variant (approximate-minimum-degree ordering on the quotient graph + up-looking LDL': the AMD + QDLDL path), 1 thread, random LPs "
                        "m = 2n with 50 nonzeros per column (the shape of BASELINE.json configs[0]), 200 iterations each; every rung a "
                        "child process capped at %.0f s wall clock — observed in this run" % args.cpu_cap_s,
                "rungs": [{k_: (round(v_, 3) if isinstance(v_, float) else v_) for k_, v_ in r.items() if k_ != "mode"} for r in rungs],
                "largest_finished": ({"m": done[-1]["m"], "n": done[-1]["n"], "value": round(done[-1]["iters_per_s"], 2),
                                      "unit": "ADMM iters/s", "cores": 1, "factorization_s": round(done[-1]["factorization_s"], 2),
                                      "hip_same_workload_iters_per_s": hip_same} if done else None),
                "first_not_finished": ({"m": notdone[0]["m"], "n": notdone[0]["n"], "cap_s": args.cpu_cap_s} if notdone else None),
                "target_and_config2": ("direct infeasible here: the NUMERIC factorisation of the m=%d rung did not finish within %.0f s — fill, not "
                                       "ordering time: nnz(L) = 0.058 N^2 for this random pattern (see nnz_L of the finished rungs; x 4 per rung, "
                                       "flops x 8); config 2 (m=2e5) and the target (m=2e6) are 12x / 125x larger"
                                       % (notdone[0]["m"], args.cpu_cap_s)) if notdone else
                                      "every rung finished; config 2 (m=2e5) and the target (m=2e6) were not attempted",
            },
        }

    out["cpu_baseline"] = cpu_baseline
    out["config5_batch"] = batch_leg
    out["other_configs"] = other
    print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
