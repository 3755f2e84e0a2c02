mkdir -p gpurun_out/r4g
hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_path_lab tools/stream_path_lab.hip && ./gpurun_out/stream_path_lab 2>&1 | tee gpurun_out/r4g/stream_path_lab.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/cs_lab tools/cs_lab.hip && LAB_BASE=1 timeout 900 ./gpurun_out/cs_lab 2>&1 | head -30 | tee gpurun_out/r4g/cs_lab.txt
timeout 900 python bench.py 2>gpurun_out/r4g/bench.err | tail -1 > gpurun_out/r4g/bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4g/bench_default.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("k1_us"), d["roofline"].get("k2_us"))
print("steady", d.get("steady_window"))
print("config5", json.dumps(d.get("config5_batch"))[:900])
for o in d.get("other_configs", []):
    print(o["config"]["workload"][:40], o["value"], o.get("roofline", {}).get("frac"))
PY
for w in 8 3 2; do
SCS_HIP_CHUNK_WINDOW=$w timeout 600 python bench.py --no-cpu-baseline --no-batch --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('window $w: value', d['value'], 'steady', d['steady_window']['value'] if d.get('steady_window') else None)"
done
LINSYS=hip_dense timeout 900 python tools/batch_shard_sim.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4g/shard_sim_dense.txt
