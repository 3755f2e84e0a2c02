show() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
for o in d['other_configs'] or []:
    if 'config4' in o['config']['workload']: print('$1: config4 cold', o['value'], 'steady', o['steady_window']['value'], 'whole', o['whole_solve']['value'], o['whole_solve']['iterations'])
"; }
SCS_HIP_PSD_COOP=0 BENCH_OTHER=config4_psd timeout 600 python bench.py --no-batch --no-cpu-baseline --no-steady --workload config2_lp_soc --steps 100 --warmup 10 2>/dev/null | tail -1 | show "coop off, main(config2)+c4"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SCS_HIP_PSD_COOP=0 BENCH_OTHER=config4_psd
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/c4slow/trace -o run -- python3 bench.py --no-batch --no-cpu-baseline --no-steady --workload config2_lp_soc --steps 100 --warmup 10 > gpurun_out/c4slow.log 2>&1
python3 tools/rocpd_summary.py $(find gpurun_out/c4slow/trace -name "*.db" | head -1) 2>&1 | head -16 | cut -c1-175
