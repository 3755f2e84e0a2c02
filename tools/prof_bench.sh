#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]   (GPU box, repo root): rocprofv3 kernel trace + stats of the bench command
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/bench_$T
mkdir -p $O
timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 bench.py --no-cpu-baseline --no-batch --no-other-configs "$@" > $O/bench_trace.log 2>&1
tail -1 $O/bench_trace.log | cut -c1-600
python3 tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/summary.txt 2>&1
head -45 $O/summary.txt
