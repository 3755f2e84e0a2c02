#!/bin/bash
# how many CG steps a queued iteration's chunk is sized for (SCS_HIP_CHUNK_WINDOW = largest count of the last W solves, + 1): metric workload, 20 + 5 steps
cd $GRAFT_REPO_ROOT
for w in 3 2 1 3 2 1; do
  echo "== SCS_HIP_CHUNK_WINDOW=$w"
  SCS_HIP_CHUNK_WINDOW=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-other-configs 2>/dev/null | python -c '
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"): continue
    d = json.loads(line)
    sw = d.get("steady_window") or {}
    print("   cold 20 steps: %.1f iters/s (%.3f ms/step)   steady window: %s iters/s" % (d["value"], d["ms_per_step"], sw.get("value")))
'
done
