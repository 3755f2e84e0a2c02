mkdir -p gpurun_out/r4k
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -3 | tee gpurun_out/r4k/parity.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-batch --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value', d['value'], 'steady', d['steady_window']['value'], 'frac', r['frac'], r.get('kernel'), json.dumps(r)[:600])" | tee -a gpurun_out/r4k/bench.txt
done
