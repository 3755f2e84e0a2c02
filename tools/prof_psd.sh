#!/bin/bash
# usage: tools/prof_psd.sh <tag>   (GPU box, repo root): kernel trace + MFMA counters of K9 on 50 matrices of order 200,
# 8 calls (1 cold + 7 warm-started, 1e-3 relative perturbation between calls), split mode and one-launch mode
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1
O=gpurun_out/psd_$T
mkdir -p $O
for mode in 1 0; do
  rocprofv3 --kernel-trace --stats -d $O/trace_$mode -o run -- ./devtools/psd_run 200 50 8 1e-3 $mode > $O/trace_$mode.log 2>&1
  python3 tools/rocpd_summary.py $(find $O/trace_$mode -name "*.db" | head -1) > $O/summary_$mode.txt 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_$mode -o run -- ./devtools/psd_run 200 50 8 1e-3 $mode > $O/pmc_$mode.log 2>&1
  python3 tools/rocpd_summary.py $(find $O/pmc_$mode -name "*.db" | head -1) > $O/pmc_$mode.txt 2>&1
  echo "== mode $mode"; head -8 $O/summary_$mode.txt | cut -c1-170; cat $O/pmc_$mode.txt | cut -c1-160 | head -14
done
