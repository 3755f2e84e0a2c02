cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for L in hip_dense hip_indirect; do
O=gpurun_out/batchtrace_$L; mkdir -p $O
LINSYS=$L MAXIT=${MAXIT:-400} SCS_HIP_POOL_MB=16384 timeout 400 rocprofv3 --kernel-trace -d $O/trace -o run -- python3 tools/batch_leg.py 512 16 1 > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-330
python3 tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) | cut -c1-175 | head -34
find $O -name "*.db" -delete
done
