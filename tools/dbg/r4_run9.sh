mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -4 | tee gpurun_out/r4i/parity.txt
for sc in 2 3 3 2; do
SCS_HIP_CS_SCHED=$sc timeout 600 python bench.py --no-cpu-baseline --no-batch --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('sched $sc: value', d['value'], 'steady', d['steady_window']['value'], 'frac', r['frac'], {k:r[k] for k in r if 'us' in k or 'achieved' in k})" | tee -a gpurun_out/r4i/sched_ab.txt
done
