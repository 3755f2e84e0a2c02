"""Batch-of-problems sharding across GPUs.

The reference has no parallelism strategy at all (none; the closest is "independent
SCS instances may run concurrently", R:test/test_thread_safety.py:78-93).  A cone program does
not shard internally without a per-CG-step all-reduce, but a BATCH of independent programs
shards perfectly: problem i -> rank i mod world, no data-path collective, and ONE gather of the
padded [header | x | y | s] blocks to rank 0 at the end (RCCL over xGMI when the process group
is "nccl"; every rank sends one message straight to rank 0 — point-to-point links, no ring).
With RCCL the payload lives in HBM from the start: every solve copies its (x, y, s) device to
device from the solver's buffers into its row of the payload tensor (scs_hip_solution_to_device)
and only the 8 header scalars go up from the host — the gather reads the solutions where they are.

Inside a rank the local shard is solved as ONE grouped solve (scs.solve_batch -> scs_hip_solve_batch,
csrc/batch.hpp): equally shaped problems advance through the ADMM loop in lock step and share every
kernel launch, which is what lifts small problems off the command-queue ceiling of "one problem per
stream" (grouped=False restores that mode: `threads` solves in flight, one stream each).

    results = solve_sharded(problems)            # under torchrun, one rank per GPU
    # rank 0: list of dicts (x, y, s, info-subset) in the original order; other ranks: None

`solve_fn` is injectable so that the distributed logic is testable on CPU with gloo
(tests/test_batch_gloo.py); the default is this package's HIP backend.
"""
import numpy as np

_HDR = 8  # n, m, status_val, iter, pobj, dobj, solve_time_ms, cg_iters


def shard_indices(n_items, rank, world):
    """round-robin: problem i runs on rank i % world"""
    return list(range(rank, n_items, world))


def _make_solver(data, cone, settings):
    """settings may name the HIP linear solver ("linear_solver": HIP_INDIRECT (default) or HIP_DENSE)"""
    import scs
    settings = dict(settings)
    ls = settings.pop("linear_solver", scs.LinearSolver.HIP_INDIRECT)
    return scs.SCS(data, cone, linear_solver=ls, **settings)


def _default_solve(data, cone, settings):
    return _make_solver(data, cone, settings).solve()


def pack_result(sol, width):
    x, y, s, info = sol["x"], sol["y"], sol["s"], sol["info"]
    n, m = x.size, y.size
    row = np.zeros(width, dtype=np.float64)
    row[:_HDR] = (n, m, info["status_val"], info["iter"], info["pobj"], info["dobj"],
                  info["solve_time"], info.get("cg_iters", 0))
    row[_HDR:_HDR + n] = x
    row[_HDR + n:_HDR + n + m] = y
    row[_HDR + n + m:_HDR + n + 2 * m] = s
    return row


def unpack_result(row):
    n, m = int(row[0]), int(row[1])
    return {"x": row[_HDR:_HDR + n].copy(), "y": row[_HDR + n:_HDR + n + m].copy(),
            "s": row[_HDR + n + m:_HDR + n + 2 * m].copy(),
            "info": {"status_val": int(row[2]), "iter": int(row[3]), "pobj": float(row[4]),
                     "dobj": float(row[5]), "solve_time": float(row[6]), "cg_iters": int(row[7])}}


def _fill_row(solver, sol, row):
    """x | y | s of the solver's last solve straight from its HBM buffers into `row` (a float64 CUDA tensor slice),
    header from the host"""
    import torch
    info = sol["info"]
    n, m = sol["x"].size, sol["y"].size
    base = row.data_ptr() + 8 * _HDR
    solver._solver.solution_to_device(base, base + 8 * n, base + 8 * (n + m))
    hdr = torch.tensor([n, m, info["status_val"], info["iter"], info["pobj"], info["dobj"], info["solve_time"],
                        info.get("cg_iters", 0)], dtype=torch.float64)
    row[:_HDR].copy_(hdr)
    return info


def _solve_into_row(data, cone, settings, row):
    """default backend, device-resident result: solve, then hand the result over in HBM.  Returns the info dict."""
    solver = _make_solver(data, cone, settings)
    return _fill_row(solver, solver.solve(), row)


def _grouped_local_solve(local_problems, threads, timing=None):
    """the local shard as one grouped solve: workspaces are set up `threads` at a time (scs_init releases the GIL),
    then ONE scs.solve_batch call.  Returns (solvers, results)."""
    import time
    import scs
    if not local_problems:  # more ranks than problems: this rank only takes part in the gather
        if timing is not None:
            timing["init_s"] = timing["solve_s"] = 0.0
        return [], []
    t0 = time.perf_counter()

    def make(p):
        return _make_solver(*p)

    if threads > 1 and len(local_problems) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as pool:
            solvers = list(pool.map(make, local_problems))
    else:
        solvers = [make(p) for p in local_problems]
    t1 = time.perf_counter()
    results = scs.solve_batch(solvers)
    if timing is not None:
        timing["init_s"] = t1 - t0
        timing["solve_s"] = time.perf_counter() - t1
    return solvers, results


def solve_sharded(problems, solve_fn=None, dims=None, device=None, threads=1, grouped=True, timing=None):
    """problems: list of (data, cone, settings) — every rank passes the same list (or at least the
    same length and `dims` = [(n, m), ...]); only its own shard is touched.  Returns the ordered
    result list on rank 0 and None elsewhere.  Works without torch.distributed (world = 1).

    Default backend: the local shard is one grouped solve (see the module docstring); `threads` workspaces are
    set up concurrently.  grouped=False (or an injected solve_fn): threads > 1 solves that many problems of the
    local shard concurrently — every SCS instance owns its HIP stream and lock and the backend releases the GIL for
    the whole solve (the reference's contract for independent instances, R:test/test_thread_safety.py:78-93):
    "one problem per stream".  timing: optional dict, receives {"init_s", "solve_s"} of the local shard (grouped mode)."""
    import torch
    import torch.distributed as dist

    device_rows = solve_fn is None  # the product backend can hand its result over in HBM
    solve_fn = solve_fn or _default_solve
    use_dist = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if use_dist else 0
    world = dist.get_world_size() if use_dist else 1
    N = len(problems)
    if dims is None:
        dims = [(len(p[0]["c"]), len(p[0]["b"])) for p in problems]
    width = _HDR + max(n + 2 * m for n, m in dims)
    per_rank = (N + world - 1) // world
    mine = shard_indices(N, rank, world)
    if use_dist and device_rows and dist.get_backend() == "nccl":
        # RCCL: the payload is a device tensor from the start; the solves fill their rows device to device
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        payload = torch.zeros((per_rank, width), dtype=torch.float64, device=device)
        torch.cuda.synchronize(device)
        if grouped:
            import time
            t0 = time.perf_counter()
            solvers, results = _grouped_local_solve([problems[i] for i in mine], threads, timing)
            for slot, (sv, res) in enumerate(zip(solvers, results)):
                _fill_row(sv, res, payload[slot])
            if timing is not None:
                timing["local_s"] = time.perf_counter() - t0
            del solvers
        elif threads > 1 and len(mine) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=threads) as pool:
                futs = [pool.submit(_solve_into_row, *problems[i], payload[slot]) for slot, i in enumerate(mine)]
                for f in futs:
                    f.result()
        else:
            for slot, i in enumerate(mine):
                _solve_into_row(*problems[i], payload[slot])
        torch.cuda.synchronize(device)
        bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
        dist.gather(payload, bufs, dst=0)  # the single collective of the whole batch, HBM to HBM
        if rank != 0:
            return None
        out = [None] * N
        for r in range(world):
            arr = bufs[r].cpu().numpy()
            for slot, i in enumerate(shard_indices(N, r, world)):
                out[i] = unpack_result(arr[slot])
        return out
    block = np.zeros((per_rank, width), dtype=np.float64)
    if device_rows and grouped:
        import time
        t0 = time.perf_counter()
        solvers, results = _grouped_local_solve([problems[i] for i in mine], threads, timing)
        t1 = time.perf_counter()
        for slot, res in enumerate(results):
            block[slot] = pack_result(res, width)
        if timing is not None:
            timing["local_s"] = time.perf_counter() - t0
            timing["pack_s"] = time.perf_counter() - t1
        del solvers
    elif threads > 1 and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as pool:
            futs = [pool.submit(solve_fn, *problems[i]) for i in mine]
            for slot, f in enumerate(futs):
                block[slot] = pack_result(f.result(), width)
    else:
        for slot, i in enumerate(mine):
            data, cone, settings = problems[i]
            block[slot] = pack_result(solve_fn(data, cone, settings), width)
    if not use_dist:
        return [unpack_result(block[slot]) for slot in range(len(mine))]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    payload = torch.from_numpy(block).to(device)
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bufs, dst=0)  # the single collective of the whole batch
    if rank != 0:
        return None
    out = [None] * N
    for r in range(world):
        arr = bufs[r].cpu().numpy()
        for slot, i in enumerate(shard_indices(N, r, world)):
            out[i] = unpack_result(arr[slot])
    return out
