#!/usr/bin/env python3
"""Timeline of a rocprofv3 (rocpd sqlite) kernel trace: per kernel its duration and the idle gap before it."""
import sqlite3, sys, re
db = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0; count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"^void\s+", "", n).replace("scship::", "")
    return n.split("(")[0][:46]
tot_busy = tot_gap = 0
prev_end = None
for i, (name, st, en) in enumerate(rows):
    gap = (st - prev_end) if prev_end is not None else 0
    if skip <= i < skip + count:
        print("%6d %-46s dur %7.2f us   gap before %7.2f us" % (i, short(name), (en - st) / 1e3, gap / 1e3))
    if i >= skip:
        tot_busy += en - st; tot_gap += max(gap, 0)
    prev_end = max(prev_end or 0, en)
print("from kernel %d on: busy %.1f ms, idle between kernels %.1f ms (%d kernels)" % (skip, tot_busy / 1e6, tot_gap / 1e6, len(rows) - skip))
