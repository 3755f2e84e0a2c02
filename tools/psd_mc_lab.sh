#!/bin/bash
# A/B of the split-mode sweeps: one workgroup per matrix vs G workgroups per matrix (k_psd_sweep_mc), bitwise comparison included.
#   gpurun -- bash tools/psd_mc_lab.sh
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/psd_lab "$ROOT/tools/psd_lab.hip" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPSD_MC_FORCE_WT -o /tmp/psd_lab_wt "$ROOT/tools/psd_lab.hip" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPSD_PROFILE=1 -o /tmp/psd_lab_p "$ROOT/tools/psd_lab.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
show() { grep -E "^ +[0-9]|bitwise|TIMEOUT" | sed -e "s/|.*//" | paste -sd" " | sed -e "s/ \+/ /g"; }
{
echo "=== phase timers, order 200 x 50, G 4, look-ahead"
timeout 120 /tmp/psd_lab_p 200 50 3 1e-2 4 0 | grep -E "mc member 0|pivot ahead|^ +[0-9]" | cut -c1-160
echo "=== phase timers, order 200 x 50, G 4, two barriers per step"
PSD_LAB_LA0=1 timeout 120 /tmp/psd_lab_p 200 50 3 1e-2 4 0 | grep -E "mc member 0|pivot ahead|^ +[0-9]" | cut -c1-160
for G in 4; do
echo "=== order 200 x 50, G $G with look-ahead, then one workgroup per matrix: us per call, bits"
timeout 120 /tmp/psd_lab 200 50 8 1e-2 $G 1 | show
done
echo "=== G 4 without look-ahead"
PSD_LAB_LA0=1 timeout 120 /tmp/psd_lab 200 50 8 1e-2 4 1 | show
echo "=== write-through stores forced, G 4"
timeout 120 /tmp/psd_lab_wt 200 50 8 1e-2 4 1 | show
echo "=== order 500 x 6, G 8"
timeout 120 /tmp/psd_lab 500 6 5 1e-2 8 1 | show
echo "=== order 1000 x 2, G 8"
timeout 300 /tmp/psd_lab 1000 2 3 1e-2 8 1 | show
echo "=== order 100 x 20, G 6 (look-ahead off: H > 8 G does not hold? H = 7)"
timeout 120 /tmp/psd_lab 100 20 5 1e-2 6 1 | show
echo "=== order 100 x 20, G 2"
timeout 120 /tmp/psd_lab 100 20 5 1e-2 2 1 | show
echo "=== order 40 x 30, G 3"
timeout 120 /tmp/psd_lab 40 30 5 1e-2 3 1 | show
} > "$ROOT/gpurun_out/psd_mc_lab.txt" 2>&1
