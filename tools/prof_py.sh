#!/bin/bash
# usage: tools/prof_py.sh <tag> <script.py> [args...]   (GPU box, repo root): kernel trace + summary + timeline of any script
# PROF_TIMEOUT (seconds, default 300) bounds the profiled run: a crashed rocprofv3 child otherwise waits for gpurun's limit
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/py_$T
mkdir -p $O
timeout ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 "$@" > $O/run.log 2>&1
tail -1 $O/run.log | cut -c1-300
DB=$(find $O/trace -name "*.db" | head -1)
python3 tools/rocpd_summary.py $DB > $O/summary.txt 2>&1
head -22 $O/summary.txt | cut -c1-175
