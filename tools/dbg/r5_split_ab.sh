#!/bin/bash
# the 2-way column split of K1 (A) with the in-kernel ticket combine, re-measured on the ROUND-4 kernel
# (k_spmv_cs_il<.., 6>: stream loads before the barrier), against the shipped layout (A unsplit, A' in two partial vectors), same box.
#   bash tools/dbg/r5_split_ab.sh > gpurun_out/r05_split_ab.txt
cd $GRAFT_REPO_ROOT
run() {
  echo "== $1"
  shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-other-configs --no-steady 2>/dev/null | python -c '
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"): continue
    d = json.loads(line)
    r = d.get("roofline", {})
    print("   iters/s %.1f   ms/step %.3f   K1 %.2f us (%.4f)  K2 %.2f us (%.4f)   line frac %.4f" % (d["value"], d["ms_per_step"], 1e3 * r["k1"]["avg_ms"], r["k1"]["frac"], 1e3 * r["k2"]["avg_ms"], r["k2"]["frac"], r.get("frac", 0)))
'
}
run "shipped: A unsplit (rows per workgroup 7.8 k), A' split in two partial vectors" X=1
run "same again (box noise)" X=1
run "A split in two + in-kernel combine, A' split in two + in-kernel combine" SCS_HIP_CS_COMBINE=1 SCS_HIP_CS_SPLIT_A=2 SCS_HIP_CS_SPLIT_AT=2
run "A split in two + combine, A' split in four + combine" SCS_HIP_CS_COMBINE=1 SCS_HIP_CS_SPLIT_A=2 SCS_HIP_CS_SPLIT_AT=4
run "A unsplit, A' split in two + in-kernel combine (no partial vectors)" SCS_HIP_CS_COMBINE=1 SCS_HIP_CS_SPLIT_A=1 SCS_HIP_CS_SPLIT_AT=2
run "round-2/3 schedule, shipped layout (SCS_HIP_CS_SCHED=2)" SCS_HIP_CS_SCHED=2
