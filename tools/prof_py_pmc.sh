#!/bin/bash
# usage: tools/prof_py_pmc.sh <tag> "<counters of one pass>|<counters of the next pass>|..." <script.py> [args...]   (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
G=$1; shift
O=gpurun_out/pmc_$T
mkdir -p $O
IFS='|' read -ra GR <<< "$G"
for c in "${GR[@]}"; do
  d=$O/$(echo $c | tr ' ' '_')
  timeout ${PROF_TIMEOUT:-200} rocprofv3 --pmc $c -d $d -o run -- python3 "$@" > $d.log 2>&1
  python3 tools/rocpd_summary.py $(find $d -name "*.db" | head -1) | grep -E "^==|${PMC_FILTER:-.}"
done
