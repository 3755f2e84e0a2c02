# A/B of the workgroup size of k_psd_gemm (tasks of 2 x 4 tiles per wavefront): balance over the CUs vs waves per workgroup
cd /tmp; export TMPDIR=/tmp
for T in 64 128 256; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPSD_GEMM_THREADS=$T -o /tmp/psd_lab_$T $GRAFT_REPO_ROOT/tools/psd_lab.hip || exit 1
rm -rf /tmp/pl; PSD_LAB_PLAIN=1 rocprofv3 --kernel-trace --stats -d /tmp/pl -o t -- /tmp/psd_lab_$T 200 50 6 1e-2 4 0 > /dev/null 2>&1
echo "threads $T"; python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/pl -name "*.db" | head -1) | grep -E "gemm" | cut -c1-170
done
