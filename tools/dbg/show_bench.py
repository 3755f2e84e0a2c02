import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "steady", (d.get("steady_window") or {}).get("value"))
r = d["roofline"]; print("k1", r.get("k1"), "k2", r.get("k2"))
cb = d.get("cpu_baseline")
if cb:
    print("cpu 1 thread", cb["value"], "| multi", {k: (cb.get("multi_core") or {}).get(k) for k in ("value", "cores", "cpus_this_process_may_use", "thread_sweep_iters_per_s")})
    if cb.get("direct_ldl"):
        print("ldl rungs", [(r_["m"], r_.get("iters_per_s"), r_.get("nnz_L"), r_.get("factorization_s"), r_["finished"]) for r_ in cb["direct_ldl"]["rungs"]], cb["direct_ldl"]["largest_finished"])
b = d.get("config5_batch")
if b:
    print("config5", b["value"], b["wall_s"], b["rank0_phases_s"], b["linear_solver"], b["solved"])
for o in d.get("other_configs") or []:
    print(o["config"]["workload"][:44], o["value"], o["roofline"]["frac"], (o.get("steady_window") or {}).get("value"), o.get("whole_solve"))
