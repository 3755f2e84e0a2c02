#!/bin/bash
# usage: tools/prof_bench_pmc.sh <tag> [bench args...]  (GPU box, repo root): counters of the bench's own K1 / K2 launches,
# one rocprofv3 --pmc pass per group (as MI355X_MICROARCH.md prescribes: no trace domains next to --pmc); launches that
# returned at once on the device-side convergence flag are excluded by tools/rocpd_summary.py --executed (value > half of
# the kernel's maximum).  Groups: HBM-side traffic (FETCH_SIZE, WRITE_SIZE, TCC hit / miss), then what bounds the
# kernels upstream of HBM — texture addresser busy cycles, L1 -> L2 requests, stalls — against the busy cycles.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/benchpmc_$T
mkdir -p $O
GROUPS_LIST=${PMC_GROUPS:-"FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|GRBM_GUI_ACTIVE SQ_BUSY_CYCLES|TA_TA_BUSY_sum TA_BUSY_avr|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum|TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum|TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum|SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"}
IFS='|' read -ra GR <<< "$GROUPS_LIST"
for c in "${GR[@]}"; do
  d=$O/$(echo $c | tr ' ' '_')
  timeout ${PROF_TIMEOUT:-240} rocprofv3 --pmc $c -d $d -o run -- python3 bench.py --no-cpu-baseline --no-batch --no-steady --no-other-configs --steps 20 --warmup 2 "$@" > $d.log 2>&1
  python3 tools/rocpd_summary.py --executed $(find $d -name "*.db" | head -1) | grep -E "^==|k_spmv_cs_il<Epi(DivR|Gp),"
done
