# config-5 batch leg under different HIP hardware-queue counts / in-flight problem counts (GPU box)
for q in ${QUEUES:-1 2}; do for th in ${THREADS:-16}; do
echo "GPU_MAX_HW_QUEUES=$q threads=$th: $(GPU_MAX_HW_QUEUES=$q python bench.py --no-steady --no-cpu-baseline --steps 2 --warmup 1 --batch-threads $th 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config5_batch']['value'], d['config5_batch']['wall_s'], d['config5_batch']['total_iters'])")"
done; done
