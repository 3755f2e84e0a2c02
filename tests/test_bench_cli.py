"""bench.py's command-line contract: `--gpus N` really runs N ranks (the process becomes the launcher when no
launcher started it), and a world size that contradicts --gpus is an error rather than a silently wrong line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_world_size_mismatch_is_refused():
    """no GPU needed: the check runs before anything is imported"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=3" in r.stderr and "--gpus 2" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_gpus_2_spawns_two_ranks_on_one_gpu_box():
    """`python bench.py --gpus 2` with no launcher: two ranks are started (torch.distributed.run), here both on device
    0 with the gloo backend (a 1-GPU box); rank 0 prints the one JSON line with n_gpus = 2 = the world size it saw,
    the batch leg covers every problem exactly once across the ranks."""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "2", "--workload", "config2_lp_soc",
           "--dist-backend", "gloo", "--force-device", "0", "--batch-problems", "12", "--batch-threads", "4", "--no-other-configs"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["config"]["world_size_seen"] == 2 and out["config"]["backend"] == "gloo"
    assert out["steps"] == 20 and out["config"]["admm_iters_timed"] == 20
    assert out["value"] > 0 and abs(out["value"] - 2 * 20 / (out["ms_per_step"] * 20 * 1e-3)) < 1e-2 * out["value"]
    assert out["cpu_baseline"] is None  # N = 1 only
    assert out["steady_window"]["aa_calls_in_window"] == 12 and out["steady_window"]["value"] > 0
    b = out["config5_batch"]
    assert b["n_gpus"] == 2 and b["problems"] == 12 and b["solved"] == 12 and b["value"] > 0
    assert out["config"]["gather_ms"] is not None


@pytest.mark.gpu
def test_gpus_8_dry_run_on_one_gpu_box():
    """the plumbing of the driver's 8-GPU run, dry — 8 ranks (torch.distributed.run, gloo) on the ONE GPU of the
    box: the line says n_gpus = world size = 8, every problem of the batch is solved exactly once across the ranks (24 = 3 per
    rank), the gather happened, and a rank sets up CPU quota // 8 (>= 1) workspaces at a time (BASELINE config 5)."""
    cmd = [sys.executable, BENCH, "--gpus", "8", "--steps", "10", "--warmup", "1", "--workload", "small_lp_soc", "--no-steady",
           "--dist-backend", "gloo", "--force-device", "0", "--batch-problems", "24", "--no-other-configs"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 8 and out["config"]["world_size_seen"] == 8 and out["config"]["backend"] == "gloo"
    assert out["steps"] == 10 and out["scaling"] == "weak" and out["value"] > 0
    assert abs(out["value"] - 8 * 10 / (out["ms_per_step"] * 10 * 1e-3)) < 1e-2 * out["value"]   # whole-job rate: 8 instances
    assert out["config"]["gather_ms"] is not None and out["cpu_baseline"] is None
    b = out["config5_batch"]
    assert b["n_gpus"] == 8 and b["problems"] == 24 and b["solved"] == 24
    assert len(b["iterations_min_median_max"]) == 3 and b["total_iters"] >= 24 * b["iterations_min_median_max"][0]
    sys.path.insert(0, ROOT)
    import bench
    assert b["setup_threads_per_rank"] == max(1, bench.cpu_quota() // 8) >= 1
    assert b["linear_solver"] == "hip_dense" and "value_north_star_path" in b and b["value_north_star_path"] == b["value_hip_indirect"]
    assert "not north_star's indirect path" in b["value_is"]
    assert b["other_linear_solver"]["solved"] == 24


@pytest.mark.gpu
def test_single_gpu_line_has_every_leg():
    cmd = [sys.executable, BENCH, "--steps", "20", "--warmup", "2", "--workload", "config2_lp_soc", "--cpu-iters", "3",
           "--batch-problems", "8", "--batch-threads", "4", "--cpu-cap-s", "12", "--no-other-configs"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["dtype"] == "f64" and out["vs_baseline"] is None
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    ldl = cb["direct_ldl"]
    assert ldl["largest_finished"]["value"] > 0 and ldl["largest_finished"]["hip_same_workload_iters_per_s"] > 0
    assert len(ldl["rungs"]) == 3 and ldl["first_not_finished"]["cap_s"] == 12.0 and "infeasible" in ldl["target_and_config2"]
    nc = os.cpu_count()
    mc = cb["multi_core"]   # round 4: a thread sweep in one child (quarter / half / all of the logical CPUs), the best is the figure
    nq = mc["cpus_this_process_may_use"]   # cgroup quota / affinity (bench.cpu_quota): the sweep is half, all, twice of it
    assert 1 <= nq <= nc and mc["value"] > 0
    assert [t for t, _ in mc["thread_sweep_iters_per_s"]] == sorted({max(1, nq // 2), nq, min(nc, 2 * nq)})
    assert mc["cores"] in [t for t, _ in mc["thread_sweep_iters_per_s"]]
    assert mc["value"] == max(v for _, v in mc["thread_sweep_iters_per_s"]) or abs(mc["value"] - max(v for _, v in mc["thread_sweep_iters_per_s"])) < 1e-2
    assert out["config5_batch"]["linear_solver"] == "hip_dense"
    assert out["config5_batch"]["value_north_star_path"] == out["config5_batch"]["value_hip_indirect"] > 0   # the indirect PCG figure, labelled
    # every member's objective against the optimum its generator constructed (default settings: residuals at 1e-4, the objective ~1e-3)
    assert out["config5_batch"]["objective_checked"] == 8 and out["config5_batch"]["objective_max_rel_err_vs_constructed_optimum"] < 5e-3
    assert out["config5_batch"]["other_linear_solver"]["objective_max_rel_err_vs_constructed_optimum"] < 5e-3
    assert rf["traffic"] is None and rf["traffic_source"] is None  # (no committed counter pass for this workload)
    assert out["config"]["cg_steps_per_s"] > 0 and out["config"]["ms_per_cg_step"] > 0
    assert out["steady_window"]["aa_accepted_in_window"] >= 0
    assert out["config5_batch"]["solved"] == 8 and "grouped" in out["config5_batch"]["workload"]
    assert out["other_configs"] is None


@pytest.mark.gpu
def test_other_configs_lines_and_ungrouped_batch_leg():
    """the lines of BASELINE configs 2-4 that ride along with the default run (config 3 with its box cone, config 4 with
    the MFMA roofline), and the one-problem-per-stream batch mode kept for A/B"""
    cmd = [sys.executable, BENCH, "--steps", "10", "--warmup", "1", "--workload", "small_lp_soc", "--no-cpu-baseline", "--no-steady",
           "--batch-problems", "6", "--batch-threads", "3", "--batch-ungrouped"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    oc = out["other_configs"]
    assert [o["config"]["workload"].split(":")[0] for o in oc] == ["config2_lp_soc", "config3_mixed", "config4_psd", "target_qp", "powerlaw_lp", "banded_lp"]
    k3 = oc[3]["roofline"]["k3"]   # the QP path on the bench: K3 = P p, priced against the STORED (upper) entries
    assert k3["avg_ms"] > 0 and 0 < k3["frac"] < k3["frac_streamed"] < 1 and k3["bytes"] < k3["streamed_bytes"] and k3["nnz_triu_P"] > 8e6
    assert "'bu': '99999 values'" in oc[1]["config"]["workload"] and "m=999999" in oc[1]["config"]["workload"]
    assert oc[2]["roofline"]["bound"] == "mfma" and oc[2]["roofline"]["frac"] > 0 and oc[0]["roofline"]["bound"] == "hbm"
    assert all(o["value"] > 0 and o["steps"] == o["config"]["admm_iters_timed"] for o in oc)
    assert oc[-1]["roofline"]["frac"] > oc[0]["roofline"]["frac"]  # gathers with locality: the ceiling of the decomposition
    assert oc[4]["roofline"]["bound"] == "hbm" and "row lengths" in oc[4]["config"]["why_this_line"]  # power-law rows: the pass layout is kept
    assert out["config5_batch"]["solved"] == 6 and "one problem per stream" in out["config5_batch"]["workload"]
