cd /tmp; export TMPDIR=/tmp
for T in 256 512 1024; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPSD_APPLY_THREADS=$T -o /tmp/psd_lab_$T $GRAFT_REPO_ROOT/tools/psd_lab.hip || exit 1
echo "threads $T: $(/tmp/psd_lab_$T 200 50 6 1e-2 4 0 | grep -E '^ +[0-9]' | sed -e 's/|.*//' | paste -sd' ')"
rm -rf /tmp/pl; PSD_LAB_PLAIN=1 rocprofv3 --kernel-trace --stats -d /tmp/pl -o t -- /tmp/psd_lab_$T 200 50 6 1e-2 4 0 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/pl -name "*.db" | head -1) | grep -E "apply_v" | cut -c1-170
done
