set -x
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_dense_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r4a/dense_tests.txt
cat gpurun_out/r4a/dense_tests.txt
for m in solo dense; do timeout 300 python tools/dbg/small_iter_latency.py $m 3000 2>&1 | tail -2; done | tee gpurun_out/r4a/latency.txt
