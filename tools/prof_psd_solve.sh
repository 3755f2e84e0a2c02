#!/bin/bash
# usage: tools/prof_psd_solve.sh <tag> [MAXIT]   (GPU box, repo root): kernel trace + MFMA counters of ONE config-4 solve stopped at MAXIT
# iterations (BASELINE.json configs[3]: 50 PSD cones of order 200 + l; tools/dbg/c4_prof.py) — the window past the cold start included,
# which tools/prof_psd.sh (bench.py's first 100 iterations) does not reach.  Counters in their own pass (--pmc alone).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1
export MAXIT=${2:-650}
O=gpurun_out/psd_solve_$T
mkdir -p $O
timeout ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 tools/dbg/c4_prof.py > $O/trace.log 2>&1
tail -1 $O/trace.log | cut -c1-300
python3 tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/summary.txt 2>&1
head -30 $O/summary.txt | cut -c1-175
timeout ${PROF_TIMEOUT:-300} rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o run -- python3 tools/dbg/c4_prof.py > $O/pmc.log 2>&1
tail -1 $O/pmc.log | cut -c1-300
python3 tools/rocpd_summary.py $(find $O/pmc -name "*.db" | head -1) | grep -E "^==|k_proj_psd|k_psd" | cut -c1-170
