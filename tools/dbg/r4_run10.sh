mkdir -p gpurun_out/r4j
bash tools/prof_bench.sh r04a 2>&1 | tee gpurun_out/r4j/kernel_trace_summary.txt
bash tools/prof_bench_pmc.sh r04a 2>&1 | tee gpurun_out/r4j/pmc_summary.txt
