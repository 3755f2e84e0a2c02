# per-call time of the K9 pipeline variants: one launch (MODE 0), split with one workgroup per matrix (MODE 1), split with 4 CUs per matrix
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/psd_lab $GRAFT_REPO_ROOT/tools/psd_lab.hip || exit 1
show() { grep -E "^ +[0-9]" | sed -e "s/|.*//" | paste -sd" " | sed -e "s/ \+/ /g"; }
for args in "200 50 6 1e-2 0" "200 50 6 1e-2 1" "200 50 6 1e-2 4" "200 256 5 1e-2 0" "64 256 5 1e-2 0" "500 6 4 1e-2 0"; do
  echo "args $args: $(/tmp/psd_lab $args 0 | show)"
done
