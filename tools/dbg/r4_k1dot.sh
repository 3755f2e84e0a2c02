for v in 0 1 1 0 1 0; do
SCS_HIP_K1DOT=$v timeout 600 python bench.py --no-cpu-baseline --no-batch --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('k1dot $v: value', d['value'], 'steady', d['steady_window']['value'], 'frac', r['frac'], 'k1', r['k1']['avg_ms'], 'k2', r['k2']['avg_ms'])"
done
