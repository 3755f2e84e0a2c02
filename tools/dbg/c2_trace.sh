cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2trace; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o run -- python3 bench.py --workload config2_lp_soc --no-cpu-baseline --no-batch --no-other-configs --no-steady --steps 100 --warmup 10 > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-300
DB=$(find $O/trace -name "*.db" | head -1)
python3 tools/rocpd_summary.py $DB | head -16 | cut -c1-180
python3 tools/trace_gaps.py $DB 1500 60
find $O -name "*.db" -delete
