cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/hltrace; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o run -- python3 bench.py --no-cpu-baseline --no-batch --no-other-configs --no-steady > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-200
DB=$(find $O/trace -name "*.db" | head -1)
for k in 8 12 16 20; do python3 tools/dbg/iter_timeline.py $DB $k; done
find $O -name "*.db" -delete
