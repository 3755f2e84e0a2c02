timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -q -k "psd or config4 or sdp or cs_" 2>&1 | tail -4
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('main', d['value'], d['steady_window']['value'] if d.get('steady_window') else None)
for o in d['other_configs'] or []:
    if 'config4' in o['config']['workload']: print('$1: config4 cold', o['value'], 'steady', o['steady_window']['value'], 'whole', o['whole_solve']['value'], o['whole_solve']['iterations'], 'mfma frac', o['roofline']['frac'])
"; }
BENCH_OTHER=config4_psd timeout 600 python bench.py --no-batch --no-cpu-baseline --no-steady --workload config2_lp_soc --steps 100 --warmup 10 2>/dev/null | tail -1 | show "default (ordinary launch), main(config2)+c4"
timeout 600 python bench.py --no-batch --no-cpu-baseline --no-other-configs --workload config4_psd --steps 100 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 as main: cold', d['value'], 'steady', d['steady_window']['value'], 'frac', d['roofline']['frac'])"
