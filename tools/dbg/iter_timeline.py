#!/usr/bin/env python3
"""One ADMM iteration out of a rocprofv3 kernel trace (rocpd sqlite): the kernels between the k-th and (k+1)-th k_prep, grouped (CG steps summarised)."""
import sqlite3, sys, re
db = sys.argv[1]; k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"^void\s+", "", n).replace("scship::", "")
    return n.split("(")[0][:40]
preps = [i for i, r in enumerate(rows) if short(r[0]).startswith("k_prep")]
a, b = preps[k], preps[k + 1]
t0 = rows[a][1]
print("iteration between k_prep #%d and #%d: %.1f us wall, %d kernels" % (k, k + 1, (rows[b][1] - t0) / 1e3, b - a))
agg = {}
prev_end = rows[a][1]
for name, st, en in rows[a:b]:
    s = short(name); d = (en - st) / 1e3; g = max(st - prev_end, 0) / 1e3
    real = d > 3.0
    key = (s, real)
    e = agg.setdefault(key, [0, 0.0, 0.0]); e[0] += 1; e[1] += d; e[2] += g
    prev_end = max(prev_end, en)
for (s, real), (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-40s %-9s x%3d  busy %8.1f us  (avg %6.1f)  idle before %7.1f us" % (s, "" if real else "(<3 us)", c, d, d / c, g))
print("  sum busy %.1f us, sum idle %.1f us" % (sum(v[1] for v in agg.values()), sum(v[2] for v in agg.values())))
