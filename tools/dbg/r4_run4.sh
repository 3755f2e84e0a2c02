mkdir -p gpurun_out/r4d
python tools/dbg/dense_gemv_err.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4d/gemv_err.txt
SCS_HIP_DENSE_GEMV=full python tools/dbg/dense_gemv_err.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/r4d/gemv_err.txt
timeout 1200 python -m pytest tests/test_dense_gpu.py -q -k "group" 2>&1 | grep -E "^E  |Error|FAILED|passed|failed" | head -60 | tee gpurun_out/r4d/dense_group_tests.txt
