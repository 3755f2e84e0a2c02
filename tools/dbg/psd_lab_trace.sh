#!/bin/bash
# usage (GPU box, repo root): tools/dbg/psd_lab_trace.sh [lab binaries ...]: kernel trace of tools/psd_lab.hip builds (split mode, 4 members, ordinary launch)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for B in "${@:-devtools/psd_lab}"; do
  O=gpurun_out/psdlab_$(basename $B); mkdir -p $O
  echo "=== $B"
  PSD_LAB_PLAIN=1 timeout 200 rocprofv3 --kernel-trace -d $O/trace -o run -- ./$B 200 50 12 1e-5 4 0 > $O/log.txt 2>&1
  python3 tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) | cut -c1-175 | grep -E "k_psd|k_proj"
  find $O -name "*.db" -delete
done
