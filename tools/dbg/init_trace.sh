#!/bin/bash
# (GPU box, repo root) kernel + copy timeline of ONE scs_init of a config-5 member: tools/dbg/init_trace.sh [dense|indirect]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/inittrace; mkdir -p $O
INIT_ONLY=1 timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace -o run -- python3 tools/dbg/init_time.py ${1:-dense} 6 > $O/log.txt 2>&1
grep "init " $O/log.txt | tail -3
python3 - <<'PY'
import sqlite3, glob, re, collections
db = glob.glob("gpurun_out/inittrace/trace/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
rows = [(s, e, re.sub(r"^void\s+", "", n).replace("scship::", "").split("(")[0][:44]) for n, s, e in con.execute("select name, start, end from kernels")]
mc = [t for t in tabs if "memory_cop" in t and not t.startswith("rocpd_info")]
for t in mc[:1]:
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % t)]
    if "start" in cols and "end" in cols:
        rows += [(s, e, "<copy %s>" % (nm or "")) for nm, s, e in con.execute("select %s, start, end from %s" % ("name" if "name" in cols else "''", t))]
rows.sort()
# the last init = everything after the last k_csr... find the last big idle gap (> 200 us) before the end-of-run solve
gaps = [(rows[i][0] - rows[i - 1][1], i) for i in range(1, len(rows))]
big = [i for g, i in gaps if g > 150000]
# take the window between the 2nd last and the last such gap that contains k_rescale_norm3
wins = list(zip([0] + big, big + [len(rows)]))
sel = [w for w in wins if any("k_rescale_norm" in r[2] for r in rows[w[0]:w[1]])][-1]
w = rows[sel[0]:sel[1]]
print("one scs_init: %d GPU operations, first start to last end %.1f us, busy %.1f us" % (len(w), (w[-1][1] - w[0][0]) / 1e3, sum(e - s for s, e, _ in w) / 1e3))
agg = collections.OrderedDict()
for s, e, n in w:
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %-46s x%3d  %8.1f us" % (n, c, d))
PY
find $O -name "*.db" -delete
