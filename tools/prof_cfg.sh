#!/bin/bash
# usage: tools/prof_cfg.sh <tag> <devtools/configs.py names...>   (GPU box, repo root): kernel trace of a parity config
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/cfg_$T
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 devtools/configs.py "$@" > $O/run.log 2>&1
tail -2 $O/run.log | cut -c1-400
python3 tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/summary.txt 2>&1
head -30 $O/summary.txt
