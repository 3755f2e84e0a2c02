mkdir -p gpurun_out/r4final
timeout 1200 python bench.py 2>gpurun_out/r4final/bench.err | tail -1 > gpurun_out/r4final/bench_default.json
python tools/dbg/show_bench.py gpurun_out/r4final/bench_default.json
bash tools/prof_bench.sh r04final > gpurun_out/r4final/kernel_trace_summary.txt 2>&1
head -8 gpurun_out/r4final/kernel_trace_summary.txt | cut -c1-200
