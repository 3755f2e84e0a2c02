"""per solve of a rocprofv3 kernel trace (rocpd sqlite): span, busy time, gap statistics between consecutive kernels.
usage: python tools/dbg/trace_gaps.py run_results.db  — solves are delimited by k_prep launches more than 5 ms apart"""
import sqlite3, sys, re
import numpy as np
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
st = np.array([r[1] for r in rows], dtype=np.int64); en = np.array([r[2] for r in rows], dtype=np.int64)
cut = [0] + [i + 1 for i in range(len(rows) - 1) if st[i + 1] - en[i] > 5_000_000] + [len(rows)]
for a, b in zip(cut, cut[1:]):
    if b - a < 500: continue
    dur = en[a:b] - st[a:b]; gap = st[a + 1:b] - en[a:b - 1]
    names = {}
    for r in rows[a:b]:
        k = re.sub(r"\(.*", "", r[0]).replace("void scship::", "")[:24]
        names.setdefault(k, []).append(r[2] - r[1])
    top = sorted(names.items(), key=lambda kv: -sum(kv[1]))[:3]
    print("kernels %6d span %8.2f ms busy %8.2f ms  gaps: median %.1f us p90 %.1f us max %.1f us  | %s" % (
        b - a, (en[b - 1] - st[a]) / 1e6, dur.sum() / 1e6, np.median(gap) / 1e3, np.percentile(gap, 90) / 1e3, gap.max() / 1e3,
        ", ".join("%s n=%d avg %.1f us" % (k, len(v), np.mean(v) / 1e3) for k, v in top)))
