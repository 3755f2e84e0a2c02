set -x
mkdir -p gpurun_out/r4b
timeout 1200 python -m pytest tests/test_dense_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r4b/dense_tests.txt
cat gpurun_out/r4b/dense_tests.txt
timeout 600 python -m pytest tests/test_group_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r4b/group_tests.txt
for m in dense dense_group2; do timeout 300 python tools/dbg/small_iter_latency.py $m 3000 2>&1 | tail -1; done | tee gpurun_out/r4b/latency.txt
LINSYS=hip_dense SCS_HIP_GROUP_STATS=1 timeout 600 python tools/batch_leg.py 512 16 1 2>&1 | tail -40 | tee gpurun_out/r4b/batch_dense.txt
timeout 600 python tools/batch_leg.py 512 16 1 2>&1 | tail -3 | tee gpurun_out/r4b/batch_indirect.txt
