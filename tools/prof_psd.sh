#!/bin/bash
# usage: tools/prof_psd.sh <tag>   (GPU box, repo root): kernel trace + MFMA counters of the PSD-heavy bench workload
# (BASELINE.json configs[3]: 50 matrices of order 200 + l): bench.py --workload config4_psd, K9 = k_proj_psd / k_psd_gemm / k_psd_apply_v
cd /tmp && export TMPDIR=/tmp
# ordinary launch of the multi-CU sweep kernel: rocprofv3 7.2 segfaults at exit after a cooperative launch (same kernel, same grid)
export SCS_HIP_PSD_COOP=0
cd $GRAFT_REPO_ROOT
T=$1
O=gpurun_out/psd_$T
mkdir -p $O
ARGS="--workload config4_psd --steps 100 --warmup 5 --no-cpu-baseline --no-batch --no-steady --no-other-configs"
timeout ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 bench.py $ARGS > $O/trace.log 2>&1
tail -1 $O/trace.log | cut -c1-1500
python3 tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/summary.txt 2>&1
head -16 $O/summary.txt | cut -c1-170
timeout ${PROF_TIMEOUT:-300} rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o run -- python3 bench.py $ARGS > $O/pmc.log 2>&1
python3 tools/rocpd_summary.py $(find $O/pmc -name "*.db" | head -1) | grep -E "^==|k_proj_psd|k_psd|k_spmv" | cut -c1-170
