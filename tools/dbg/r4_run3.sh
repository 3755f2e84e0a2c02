set -x
mkdir -p gpurun_out/r4c
timeout 300 python tools/dbg/dense_scale_dbg.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c/scale_dbg.txt
timeout 1200 python -m pytest tests/test_dense_gpu.py -q 2>&1 | tail -25 > gpurun_out/r4c/dense_tests.txt
cat gpurun_out/r4c/dense_tests.txt
for m in dense dense_group2; do timeout 300 python tools/dbg/small_iter_latency.py $m 3000 2>&1 | tail -1; done | tee gpurun_out/r4c/latency.txt
LINSYS=hip_dense SCS_HIP_GROUP_STATS=1 timeout 600 python tools/batch_leg.py 512 16 1 2>&1 | grep -v "iteration  [2-9][0-9][0-9][0-9]\|iteration   [2-9]" | tail -30 | tee gpurun_out/r4c/batch_dense.txt
LINSYS=hip_dense SCS_HIP_DENSE_GEMV=full timeout 600 python tools/batch_leg.py 512 16 1 2>&1 | tail -1 | tee gpurun_out/r4c/batch_dense_fullgemv.txt
