for lib in "" build/ab/libscs_hip_ve8.so build/ab/libscs_hip_ve16.so "" build/ab/libscs_hip_ve8.so; do
SCS_HIP_LIB=${lib:+$PWD/$lib} timeout 600 python bench.py --no-cpu-baseline --no-batch --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib ${lib:-default}: value', d['value'], 'steady', d['steady_window']['value'])"
done
for lib in "" build/ab/libscs_hip_ve8.so; do
SCS_HIP_LIB=${lib:+$PWD/$lib} timeout 600 python bench.py --no-cpu-baseline --no-batch --no-other-configs --no-steady --workload config2_lp_soc --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 lib ${lib:-default}: value', d['value'])"
done
