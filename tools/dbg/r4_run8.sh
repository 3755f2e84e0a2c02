mkdir -p gpurun_out/r4h
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/cs_lab tools/cs_lab.hip 2>/dev/null && LAB_BASE=1 LAB_SKIP_COMBINE=1 timeout 900 ./gpurun_out/cs_lab 2>&1 | tee gpurun_out/r4h/cs_lab.txt
