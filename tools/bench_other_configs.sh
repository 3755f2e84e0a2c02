#!/bin/bash
# bench lines of the non-metric BASELINE configs -> gpurun_out/other_configs.jsonl   (GPU box, repo root)
O=gpurun_out/other_configs.jsonl
mkdir -p gpurun_out; : > $O
run() { echo "# bench.py $*" >> $O; python bench.py "$@" --no-cpu-baseline --no-batch 2>/dev/null | tail -1 >> $O; }
run --workload config2_lp_soc --steps 100 --warmup 10
run --workload config3_mixed --steps 100 --warmup 10
run --workload config4_psd --steps 100 --warmup 5 --no-steady
run --workload config4_psd --steps 300 --warmup 5 --no-steady
run --workload config4_psd --no-steady
