#!/bin/bash
# usage: tools/prof_bench_pmc.sh <tag> [bench args...]  (GPU box, repo root): HBM-side traffic counters of the bench's own K1 / K2
# launches (separate --pmc passes, as MI355X_MICROARCH.md prescribes); launches that returned at once on the device-side
# convergence flag are excluded by tools/rocpd_summary.py --executed (value > half of the kernel's maximum)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/benchpmc_$T
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d $d -o run -- python3 bench.py --no-cpu-baseline --no-batch --no-steady --steps 20 --warmup 2 "$@" > $d.log 2>&1
  python3 tools/rocpd_summary.py --executed $(find $d -name "*.db" | head -1) | grep -E "^==|k_spmv_cs|k_epi_finish|k_spmv_slab|k_spmv_stream"
done
